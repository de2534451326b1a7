"""Plain-PyTorch fp32 CPU restatement of the floating-point part of the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Every function takes a flat ``state_dict``-style mapping with the reference's own
parameter names and NCHW fp32 tensors, exactly like the
reference modules, and cites the reference lines it restates:

  d2z: = path inside /root/reference/detectron2.7z (the vendored, modified Detectron2)
  ref: = path under /root/reference/

Parity pinning: the reference ships no tests or golden vectors, so
this restatement is pinned by executing the reference's own files in this container
(``oracle/refrun/gen_golden.py``) on seeded inputs and committing the outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against them.
torchvision (nms / roi_align) is an un-vendored dependency: see ref_decode.c.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Mapping[str, Tensor]

# d2z:modeling/backbone/vovnet.py:28-96 -- the stage tables (only non-depthwise bodies).
VOVNET_SPECS = {
    "V-19-slim-eSE": dict(stem=(64, 64, 128), conv=(64, 80, 96, 112), out=(112, 256, 384, 512),
                          layers=3, blocks=(1, 1, 1, 1)),
    "V-19-eSE": dict(stem=(64, 64, 128), conv=(128, 160, 192, 224), out=(256, 512, 768, 1024),
                     layers=3, blocks=(1, 1, 1, 1)),
    "V-39-eSE": dict(stem=(64, 64, 128), conv=(128, 160, 192, 224), out=(256, 512, 768, 1024),
                     layers=5, blocks=(1, 1, 2, 2)),
    "V-57-eSE": dict(stem=(64, 64, 128), conv=(128, 160, 192, 224), out=(256, 512, 768, 1024),
                     layers=5, blocks=(1, 1, 4, 3)),
    "V-99-eSE": dict(stem=(64, 64, 128), conv=(128, 160, 192, 224), out=(256, 512, 768, 1024),
                     layers=5, blocks=(1, 3, 9, 3)),
}

# Operand precision of the dense convolutions.  "fp32" is the reference.  "bf16" restates include/ore_hip.h's ORE_CONV_BF16 mode
# (BASELINE configs[4]; the reference has no reduced-precision path, so this mode is pinned by construction only): both operands of
# every MFMA convolution -- all VoVNet/FPN convs except stem_1, conv3, the CenterNet head convs -- are rounded to bf16 (nearest
# even) at the point where they enter the convolution; accumulation, FrozenBN, eSE, GroupNorm, the depthwise correlation and
# everything after the head stay fp32.
# "bf16s" restates ORE_CONV_BF16S, the bf16 STORAGE mode of an engine: every activation that travels between two kernels is a bf16 tensor
# -- the producer rounds its fp32 result once, nearest even (`_st`) --, conv weights are bf16; stem_1's arithmetic, FrozenBN / bias, the
# eSE pool and gate, GroupNorm statistics, the depthwise correlation's arithmetic and the head OUTPUTS stay fp32.  Two folds of the
# product are part of the mode's definition: the eSE gate reaches the max-pool as round(max(x) * g) (= round(max(x * g)), g >= 0) and the
# FPN lateral as a weight, round(W * g) (the pair rides on the gated tensor: `_ore_gated`).
_OPERANDS = "fp32"
_LINEARS = False
FROZEN_STAGES = 2      # stage index k <= FROZEN_STAGES + 1 is frozen: stem + stage2 + stage3 (FREEZE_AT = 3, d2z vovnet.py:455-468)


class _frozen_storage:
    """Scope of the frozen stages inside the training form of the bf16 mode: storage rounding on, custom-backward convs off (nothing
    in there has a gradient)."""

    def __enter__(self):
        global _OPERANDS, _LINEARS
        self.saved = (_OPERANDS, _LINEARS)
        if _OPERANDS == "bf16" and _LINEARS:
            _OPERANDS, _LINEARS = "bf16s", False
        return self

    def __exit__(self, *exc):
        global _OPERANDS, _LINEARS
        _OPERANDS, _LINEARS = self.saved


class operand_precision:
    """with operand_precision("bf16"): ... -- scoped switch of the dense-conv operand rounding.
    train=True is the TRAINING form of the mode (tests/test_hip_bf16.py, 5-shot iteration): (i) the Linear / 1x1 layers that the
    product's training forward also runs on the MFMA conv kernel -- SM_Block's Linears, the second stage's DSA convs, fc1 and the
    predictors -- round their operands too; (ii) the backward of every such layer rounds ITS operands: dY and W for the data
    gradient, X and dY for the weight gradient (bias gradients and everything element-wise stay fp32); (iii) round 4: the FROZEN
    stages of the backbone (stem, stage 2, stage 3 at FREEZE_AT = 3 -- 70 image-equivalents of a bs-16 step, no gradients) run in the
    STORAGE form: every map between two of their kernels is a bf16 tensor (`_st`), the eSE gate reaches the max-pool as
    round(max(x) * g); what leaves them -- the gated stage-3 map (bf16 value x fp32 gate, an fp32 tensor) and stage 4's pooled input
    (a bf16 map) -- is consumed by the trainable layers in their operand-rounding form."""

    def __init__(self, mode: str, train: bool = False):
        assert mode in ("fp32", "bf16", "bf16s")
        self.mode, self.train = mode, train

    def __enter__(self):
        global _OPERANDS, _LINEARS
        self.saved, _OPERANDS, _LINEARS = (_OPERANDS, _LINEARS), self.mode, self.train and self.mode == "bf16"

    def __exit__(self, *exc):
        global _OPERANDS, _LINEARS
        _OPERANDS, _LINEARS = self.saved


def _rnd(t: Tensor) -> Tensor:
    return t.bfloat16().float() if _OPERANDS in ("bf16", "bf16s") else t


def _st(t: Tensor) -> Tensor:
    """A tensor as it is STORED between two kernels: bf16 in the storage mode, untouched otherwise."""
    return t.bfloat16().float() if _OPERANDS == "bf16s" else t




class _Bf16ConvFn(torch.autograd.Function):
    """conv2d on bf16-rounded operands whose backward rounds its operands as well (the product's data- and weight-gradient kernels
    in ORE_CONV_BF16 mode); products exact in fp32, accumulation fp32."""

    @staticmethod
    def forward(ctx, x, w, stride, padding):
        xr, wr = x.bfloat16().float(), w.bfloat16().float()
        ctx.save_for_backward(xr, wr)
        ctx.sp = (stride, padding)
        return F.conv2d(xr, wr, None, stride, padding)

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        stride, padding = ctx.sp
        dyr = dy.bfloat16().float()
        dx = torch.nn.grad.conv2d_input(xr.shape, wr, dyr, stride=stride, padding=padding) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv2d_weight(xr, wr.shape, dyr, stride=stride, padding=padding) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None


def dense_conv(x: Tensor, w: Tensor, b=None, stride: int = 1, padding: int = 0) -> Tensor:
    """F.conv2d with the operand rounding of the current mode (bias and accumulation in fp32)."""
    if _LINEARS and (x.requires_grad or w.requires_grad):
        y = _Bf16ConvFn.apply(x, w, stride, padding)
        return y if b is None else y + b.view(1, -1, 1, 1)
    return F.conv2d(_rnd(x), _rnd(w), b, stride, padding)


def dense_linear(x: Tensor, w: Tensor, b=None) -> Tensor:
    """F.linear; in the training form of the bf16 mode a 1x1 convolution over the rows like the product's (operand_precision)."""
    if not _LINEARS:
        return F.linear(x, w, b)
    shp = x.shape
    y = dense_conv(x.reshape(-1, shp[-1], 1, 1), w.reshape(w.shape[0], w.shape[1], 1, 1), b)
    return y.reshape(*shp[:-1], w.shape[0])


def dense_conv1x1(x: Tensor, w: Tensor, b=None) -> Tensor:
    """A 1x1 F.conv2d of the second stage: fp32 unless the training form of the bf16 mode is on."""
    return dense_conv(x, w, b) if _LINEARS else F.conv2d(x, w, b)


PIXEL_MEAN = (103.530, 116.280, 123.675)  # d2z:config/defaults.py PIXEL_MEAN (BGR)
PIXEL_STD = (1.0, 1.0, 1.0)


# --------------------------------------------------------------------------------------
# a4  preprocess (ref:fewx/modeling/fsod/fsod_cen.py:540-555, d2z:structures/image_list.py:69-121)
# --------------------------------------------------------------------------------------
def preprocess(image_chw: Tensor, size_divisibility: int = 32,
               mean: Sequence[float] = PIXEL_MEAN, std: Sequence[float] = PIXEL_STD) -> Tensor:
    """(x - mean) / std per BGR channel, then zero-pad bottom/right to a multiple of 32."""
    x = image_chw.to(torch.float32)
    m = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
    x = (x - m) / s
    h, w = x.shape[-2:]
    hp = (h + size_divisibility - 1) // size_divisibility * size_divisibility
    wp = (w + size_divisibility - 1) // size_divisibility * size_divisibility
    x = F.pad(x, (0, wp - w, 0, hp - h), value=0.0)
    return x.unsqueeze(0)


# --------------------------------------------------------------------------------------
# a2  FrozenBatchNorm2d (d2z:layers/batch_norm.py:44-66)
# --------------------------------------------------------------------------------------
def frozen_bn(x: Tensor, sd: SD, prefix: str, eps: float = 1e-5) -> Tensor:
    return F.batch_norm(x, sd[prefix + "running_mean"], sd[prefix + "running_var"],
                        sd[prefix + "weight"], sd[prefix + "bias"], training=False, eps=eps)


def conv_bn_relu(x: Tensor, sd: SD, name: str, stride: int, pad: int) -> Tensor:
    """conv (no bias) -> FrozenBN -> ReLU; d2z:modeling/backbone/vovnet.py:205-235."""
    if name.endswith("stem.stem_1"):                         # Cin = 3: fp32 in every mode
        x = F.conv2d(x, sd[name + "/conv.weight"], None, stride, pad)
    else:
        x = dense_conv(x, sd[name + "/conv.weight"], None, stride, pad)
    return _st(F.relu(frozen_bn(x, sd, name + "/norm.")))


# --------------------------------------------------------------------------------------
# a1  VoVNet (d2z:modeling/backbone/vovnet.py)
# --------------------------------------------------------------------------------------
def ese(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """eSE: x * relu6(fc(avgpool(x)) + 3) / 6   (vovnet.py:238-260)."""
    s = F.adaptive_avg_pool2d(x, 1)
    s = F.conv2d(s, sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])
    s = F.relu6(s + 3.0) / 6.0
    y = x * s
    if _OPERANDS == "bf16s":
        y._ore_gated = (x, s)          # lets fpn() fold the gate into the lateral weight; lives and dies with y (no global table)
    return y


def osa_module(x: Tensor, sd: SD, prefix: str, mod: str, n_layers: int, identity: bool) -> Tensor:
    """_OSA_module.forward (vovnet.py:310-332), non-depthwise, no DCN."""
    feats = [x]
    y = x
    for i in range(n_layers):
        y = conv_bn_relu(y, sd, f"{prefix}layers.{i}.{mod}_{i}", 1, 1)
        feats.append(y)
    y = torch.cat(feats, dim=1)
    y = conv_bn_relu(y, sd, f"{prefix}concat.{mod}_concat", 1, 0)
    y = ese(y, sd, prefix + "ese.")
    if identity:
        y = y + x
    return y


def vovnet(x: Tensor, sd: SD, prefix: str = "backbone.bottom_up.",
           body: str = "V-19-slim-eSE") -> Dict[str, Tensor]:
    """VoVNet.forward (vovnet.py:471-481): stem + stage2..stage5."""
    spec = VOVNET_SPECS[body]
    train_bf16 = _OPERANDS == "bf16" and _LINEARS             # training form of the bf16 mode: frozen stages in the storage form
    with _frozen_storage():
        x = conv_bn_relu(x, sd, prefix + "stem.stem_1", 2, 1)
        x = conv_bn_relu(x, sd, prefix + "stem.stem_2", 1, 1)
        x = conv_bn_relu(x, sd, prefix + "stem.stem_3", 2, 1)
    out = {"stem": x}
    for si in range(4):
        k = si + 2
        frozen = train_bf16 and si < FROZEN_STAGES
        pool_frozen = train_bf16 and si <= FROZEN_STAGES     # the pool in front of stage k reads stage k-1's (frozen) bf16 map
        if k != 2:  # _OSA_stage (vovnet.py:349-350)
            if pool_frozen:
                with _frozen_storage():
                    x = _st(F.max_pool2d(x, kernel_size=3, stride=2, ceil_mode=True))
            else:
                x = _st(F.max_pool2d(x, kernel_size=3, stride=2, ceil_mode=True))
        for b in range(spec["blocks"][si]):
            mod = f"OSA{k}_{b + 1}"
            if frozen:
                with _frozen_storage():
                    x = osa_module(x, sd, f"{prefix}stage{k}.{mod}.", mod, spec["layers"], identity=b > 0)
                if hasattr(x, "_ore_gated"):
                    del x._ore_gated                           # (the weight fold of the lateral belongs to the eval engine only)
            else:
                x = osa_module(x, sd, f"{prefix}stage{k}.{mod}.", mod, spec["layers"], identity=b > 0)
        out[f"stage{k}"] = x
    return out


# --------------------------------------------------------------------------------------
# a3  FPN (d2z:modeling/backbone/fpn.py:113-154), TOP_LEVELS=0, fuse "sum", bias, no norm
# --------------------------------------------------------------------------------------
def fpn(feats: Mapping[str, Tensor], sd: SD, prefix: str = "backbone.",
        in_features: Sequence[str] = ("stage3", "stage4", "stage5"),
        stages: Sequence[int] = (3, 4, 5)) -> Dict[str, Tensor]:
    res: Dict[str, Tensor] = {}
    prev = None
    for name, st in reversed(list(zip(in_features, stages))):
        if _OPERANDS == "bf16s" and hasattr(feats[name], "_ore_gated"):   # the gate rides on the lateral's weight: conv(x, round(W * g))
            xs, gs = feats[name]._ore_gated
            lat = F.conv2d(xs, _rnd(sd[f"{prefix}fpn_lateral{st}.weight"] * gs.view(1, -1, 1, 1)), sd[f"{prefix}fpn_lateral{st}.bias"])
        else:
            lat = dense_conv(feats[name], sd[f"{prefix}fpn_lateral{st}.weight"], sd[f"{prefix}fpn_lateral{st}.bias"])
        if prev is not None:
            lat = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        lat = _st(lat)
        prev = lat
        res[f"p{st}"] = _st(dense_conv(lat, sd[f"{prefix}fpn_output{st}.weight"],
                                       sd[f"{prefix}fpn_output{st}.bias"], padding=1))
    return {k: res[k] for k in sorted(res)}


def backbone_fpn(x: Tensor, sd: SD, body: str = "V-19-slim-eSE") -> Dict[str, Tensor]:
    """build_fcos_vovnet_fpn_backbone(...).forward (vovnet.py:527-555)."""
    return fpn(vovnet(x, sd, body=body), sd)


# --------------------------------------------------------------------------------------
# a5  SM_Block / MLP (ref:fewx/modeling/fsod/fsod_cen.py:573-630), eval mode (dropout = id)
# --------------------------------------------------------------------------------------
def sm_block(x: Tensor, sd: SD, prefix: str, seg_dim: int) -> Tensor:
    """x: [B,H,W,C] -> [B,H,W,C]."""
    B, H, W, C = x.shape
    S = C // seg_dim
    h = x.reshape(B, H, W, seg_dim, S).permute(0, 3, 2, 1, 4).reshape(B, seg_dim, W, H * S)
    h = dense_linear(h, sd[prefix + "mlp_h.weight"])
    h = h.reshape(B, seg_dim, W, H, S).permute(0, 3, 2, 1, 4).reshape(B, H, W, C)
    w = x.reshape(B, H, W, seg_dim, S).permute(0, 3, 1, 2, 4).reshape(B, seg_dim, H, W * S)
    w = dense_linear(w, sd[prefix + "mlp_w.weight"])
    w = w.reshape(B, seg_dim, H, W, S).permute(0, 2, 3, 1, 4).reshape(B, H, W, C)
    a = (h + w).permute(0, 3, 1, 2).flatten(2).mean(2)
    a = F.linear(a, sd[prefix + "reweighting.fc1.weight"], sd[prefix + "reweighting.fc1.bias"])   # [B,C]-sized: fp32 in every mode
    a = F.gelu(a)
    a = F.linear(a, sd[prefix + "reweighting.fc2.weight"], sd[prefix + "reweighting.fc2.bias"])
    a = a.reshape(B, C, 2).permute(2, 0, 1).softmax(0).unsqueeze(2).unsqueeze(2)
    y = w * a[0] + h * a[1]
    return dense_linear(y, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])


def support_prototype(p_feat: Tensor, sd: SD, level: int) -> Tensor:
    """Support branch for one FPN level (fsod_cen.py:216-227 train / :367-377 init_model).

    p_feat [N,128,h,w] -> AdaptiveAvgPool to (32|16|8)^2 -> NHWC -> SM_Block ->
    permute(0,3,2,1) (this swaps H and W: SURVEY Appendix C.3, preserved) -> mean over shots.
    Returns [1,128,s,s].
    """
    size = {3: 32, 4: 16, 5: 8}[level]
    x = F.adaptive_avg_pool2d(p_feat, (size, size)).permute(0, 2, 3, 1)
    x = sm_block(x, sd, f"vip_p{level}.", size).permute(0, 3, 2, 1)
    return x.mean(0, True)


# --------------------------------------------------------------------------------------
# a6  support kernels + depthwise correlation (fsod_cen.py:454-509 eval, :229-275 train)
# --------------------------------------------------------------------------------------
def support_kernels(s_pool: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """[1,C,s,s] -> k11 [C], k13 [C,3], k31 [C,3] (adaptive avg pools 1x1, 1x3, 3x1)."""
    k11 = F.adaptive_avg_pool2d(s_pool, (1, 1))[0, :, 0, 0]
    k13 = F.adaptive_avg_pool2d(s_pool, (1, 3))[0, :, 0, :]
    k31 = F.adaptive_avg_pool2d(s_pool, (3, 1))[0, :, :, 0]
    return k11.contiguous(), k13.contiguous(), k31.contiguous()


def correlation(q: Tensor, s_pool: Tensor, conv3_w: Tensor, conv3_b: Tensor) -> Tensor:
    """q [1,C,H,W], s_pool [1,C,s,s] -> relu(conv3(cat(attn, q))) [1,128,H,W]."""
    C = q.shape[1]
    k11 = F.adaptive_avg_pool2d(s_pool, (1, 1)).permute(1, 0, 2, 3)
    k13 = F.adaptive_avg_pool2d(s_pool, (1, 3)).permute(1, 0, 2, 3)
    k31 = F.adaptive_avg_pool2d(s_pool, (3, 1)).permute(1, 0, 2, 3)
    a = F.relu(F.conv2d(q, k11, padding=(0, 0), groups=C))
    a = F.relu(F.conv2d(a, k11, padding=(0, 0), groups=C))
    b = F.relu(F.conv2d(q, k13, padding=(0, 1), groups=C))
    b = F.relu(F.conv2d(b, k31, padding=(1, 0), groups=C))
    attn = _st(a + b + q)
    return _st(F.relu(dense_conv(torch.cat((attn, q), 1), conv3_w, conv3_b)))


# --------------------------------------------------------------------------------------
# a7  CenterNetHead (ref:CenterNet2/centernet/modeling/dense_heads/centernet_head.py:141-161)
#     only_proposal=True, with_agn_hm=True, NUM_BOX_CONVS=1, NORM=GN  (log:697-715)
# --------------------------------------------------------------------------------------
def centernet_head(feats: Sequence[Tensor], sd: SD,
                   prefix: str = "proposal_generator.centernet_head.") -> Tuple[List[Tensor], List[Tensor]]:
    regs, hms = [], []
    for l, x in enumerate(feats):
        t = _st(dense_conv(x, sd[prefix + "bbox_tower.0.weight"], sd[prefix + "bbox_tower.0.bias"], padding=1))
        t = F.group_norm(t, 32, sd[prefix + "bbox_tower.1.weight"], sd[prefix + "bbox_tower.1.bias"], eps=1e-5)
        t = _st(F.relu(t))
        hms.append(dense_conv(t, sd[prefix + "agn_hm.weight"], sd[prefix + "agn_hm.bias"], padding=1))
        r = dense_conv(t, sd[prefix + "bbox_pred.weight"], sd[prefix + "bbox_pred.bias"], padding=1)
        r = r * sd[prefix + f"scales.{l}.scale"]
        regs.append(F.relu(r))
    return regs, hms


# --------------------------------------------------------------------------------------
# hot path a1..a7 in one call (eval): image -> (reg, hm) per level
# --------------------------------------------------------------------------------------
def eval_dense(image_chw: Tensor, sd: SD, support: Mapping[str, Tensor],
               body: str = "V-19-slim-eSE") -> Dict[str, object]:
    """support: {'p3': [1,128,32,32], 'p4': [1,128,16,16], 'p5': [1,128,8,8]} (cached prototypes)."""
    x = preprocess(image_chw)
    feats = backbone_fpn(x, sd, body)
    pos = [correlation(feats[k], support[k], sd["conv3.weight"], sd["conv3.bias"]) for k in ("p3", "p4", "p5")]
    regs, hms = centernet_head(pos, sd)
    return {"features": feats, "pos_features": pos, "reg": regs, "hm": hms}


# --------------------------------------------------------------------------------------
# Synthetic weights: reference initialisers + non-trivial FrozenBN statistics.
# --------------------------------------------------------------------------------------
def synth_state_dict(seed: int = 0, body: str = "V-19-slim-eSE", fpn_ch: int = 128) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    spec = VOVNET_SPECS[body]
    sd: Dict[str, Tensor] = {}

    def randn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    def rand(*shape, lo=0.0, hi=1.0):
        return torch.rand(*shape, generator=g) * (hi - lo) + lo

    def conv_bn(name, cin, cout, k):
        fan_in = cin * k * k
        sd[name + "/conv.weight"] = randn(cout, cin, k, k, std=math.sqrt(2.0 / fan_in))
        sd[name + "/norm.weight"] = rand(cout, lo=0.5, hi=1.5)
        sd[name + "/norm.bias"] = randn(cout, std=0.1)
        sd[name + "/norm.running_mean"] = randn(cout, std=0.1)
        sd[name + "/norm.running_var"] = rand(cout, lo=0.5, hi=1.5)

    p = "backbone.bottom_up."
    stem = spec["stem"]
    conv_bn(p + "stem.stem_1", 3, stem[0], 3)
    conv_bn(p + "stem.stem_2", stem[0], stem[1], 3)
    conv_bn(p + "stem.stem_3", stem[1], stem[2], 3)
    cin = stem[2]
    for si in range(4):
        k = si + 2
        sc, oc = spec["conv"][si], spec["out"][si]
        for b in range(spec["blocks"][si]):
            mod = f"OSA{k}_{b + 1}"
            pre = f"{p}stage{k}.{mod}."
            c = cin
            for i in range(spec["layers"]):
                conv_bn(f"{pre}layers.{i}.{mod}_{i}", c, sc, 3)
                c = sc
            conv_bn(f"{pre}concat.{mod}_concat", cin + spec["layers"] * sc, oc, 1)
            sd[pre + "ese.fc.weight"] = randn(oc, oc, 1, 1, std=math.sqrt(1.0 / oc))
            sd[pre + "ese.fc.bias"] = randn(oc, std=0.5)
            cin = oc
    for st, c in zip((3, 4, 5), spec["out"][1:]):
        # c2_xavier_fill = kaiming_uniform_(a=1): U(-sqrt(3/fan_in), +)
        for nm, ci, kk in ((f"backbone.fpn_lateral{st}", c, 1), (f"backbone.fpn_output{st}", fpn_ch, 3)):
            bound = math.sqrt(3.0 / (ci * kk * kk))
            sd[nm + ".weight"] = rand(fpn_ch, ci, kk, kk, lo=-bound, hi=bound)
            sd[nm + ".bias"] = randn(fpn_ch, std=0.02)
    C = fpn_ch
    for lvl in (3, 4, 5):
        pre = f"vip_p{lvl}."
        b = 1.0 / math.sqrt(C)
        sd[pre + "mlp_h.weight"] = rand(C, C, lo=-b, hi=b)
        sd[pre + "mlp_w.weight"] = rand(C, C, lo=-b, hi=b)
        sd[pre + "reweighting.fc1.weight"] = rand(C // 2, C, lo=-b, hi=b)
        sd[pre + "reweighting.fc1.bias"] = rand(C // 2, lo=-b, hi=b)
        b2 = 1.0 / math.sqrt(C // 2)
        sd[pre + "reweighting.fc2.weight"] = rand(2 * C, C // 2, lo=-b2, hi=b2)
        sd[pre + "reweighting.fc2.bias"] = rand(2 * C, lo=-b2, hi=b2)
        sd[pre + "proj.weight"] = rand(C, C, lo=-b, hi=b)
        sd[pre + "proj.bias"] = rand(C, lo=-b, hi=b)
    for nm, co, ci in (("conv1", C // 2, C), ("conv2", C // 2, C), ("conv3", C, 2 * C)):
        b = 1.0 / math.sqrt(ci)
        sd[nm + ".weight"] = rand(co, ci, 1, 1, lo=-b, hi=b)
        sd[nm + ".bias"] = rand(co, lo=-b, hi=b)
    h = "proposal_generator.centernet_head."
    sd[h + "bbox_tower.0.weight"] = randn(C, C, 3, 3, std=0.01)
    sd[h + "bbox_tower.0.bias"] = torch.zeros(C)
    sd[h + "bbox_tower.1.weight"] = rand(C, lo=0.5, hi=1.5)
    sd[h + "bbox_tower.1.bias"] = randn(C, std=0.1)
    sd[h + "bbox_pred.weight"] = randn(4, C, 3, 3, std=0.01)
    sd[h + "bbox_pred.bias"] = torch.full((4,), 8.0)
    sd[h + "agn_hm.weight"] = randn(1, C, 3, 3, std=0.01)
    sd[h + "agn_hm.bias"] = torch.full((1,), -math.log(99.0))
    for l in range(3):
        sd[h + f"scales.{l}.scale"] = torch.tensor([1.0 + 0.1 * l])
    sd["pixel_mean"] = torch.tensor(PIXEL_MEAN).view(3, 1, 1)
    sd["pixel_std"] = torch.tensor(PIXEL_STD).view(3, 1, 1)
    return sd


def synth_support(seed: int = 0, C: int = 128) -> Dict[str, Tensor]:
    """Cached eval support prototypes: N(0,1)*0.1."""
    g = torch.Generator().manual_seed(seed + 1000)
    return {f"p{l}": torch.randn(1, C, s, s, generator=g) * 0.1 for l, s in ((3, 32), (4, 16), (5, 8))}


def synth_image(seed: int = 0, h: int = 640, w: int = 640) -> Tensor:
    """uint8 BGR CHW 'ore-like' texture: low-passed uniform noise in [40,200]."""
    g = torch.Generator().manual_seed(seed + 2000)
    x = torch.rand(1, 3, h // 4 + 2, w // 4 + 2, generator=g)
    x = F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False)
    x = x + 0.15 * (torch.rand(1, 3, h, w, generator=g) - 0.5)
    x = (x.clamp(0, 1) * 160 + 40).round().to(torch.uint8)
    return x[0]


# --------------------------------------------------------------------------------------
# f1  second stage, eval (CustomCascadeROIHeads): ref:fewx/modeling/fsod/fsod_roi_heads.py:404-520
# --------------------------------------------------------------------------------------
def assign_levels(boxes: Tensor, min_level: int = 3, max_level: int = 5, canonical_size: float = 224.0,
                  canonical_level: int = 4) -> Tensor:
    """d2z:modeling/poolers.py:22-58."""
    sizes = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_size + 1e-8))
    return torch.clamp(lv, min=min_level, max=max_level).to(torch.int64) - min_level


def roi_align(feat: Tensor, boxes: Tensor, scale: float, pooled: int) -> Tensor:
    """torchvision.ops.roi_align(aligned=True, sampling_ratio=0) restated from its published algorithm (torchvision 0.8.2 is an
    un-vendored dependency of the reference: d2z:layers/roi_align.py:49-65).  feat [C,H,W], boxes [R,4] -> [R,C,pooled,pooled]."""
    C, H, W = feat.shape
    f32 = feat.dtype                 # fp32 in every parity use; fp64 only for the conditioning runs of oracle/refrun/gen_golden.py
    out = torch.zeros(len(boxes), C, pooled, pooled, dtype=f32)
    for r in range(len(boxes)):
        b = boxes[r].to(f32) * scale - 0.5
        x0, y0, x1, y1 = b[0], b[1], b[2], b[3]
        rw, rh = x1 - x0, y1 - y0
        bw, bh = rw / pooled, rh / pooled
        gh, gw = int(math.ceil(float(rh) / pooled)), int(math.ceil(float(rw) / pooled))
        cnt = max(gh * gw, 1)
        if gh <= 0 or gw <= 0:
            continue
        ph = torch.arange(pooled, dtype=f32).view(-1, 1)
        iy = torch.arange(gh, dtype=f32).view(1, -1)
        ys = (y0 + ph * bh + (iy + 0.5) * bh / gh).reshape(-1)          # [pooled*gh]
        pw = torch.arange(pooled, dtype=f32).view(-1, 1)
        ix = torch.arange(gw, dtype=f32).view(1, -1)
        xs = (x0 + pw * bw + (ix + 0.5) * bw / gw).reshape(-1)          # [pooled*gw]

        def prep(v, size):
            oob = (v < -1.0) | (v > size)
            v = torch.clamp(v, min=0.0)
            lo = v.to(torch.int64)
            hi = lo + 1
            edge = lo >= size - 1
            lo = torch.where(edge, torch.full_like(lo, size - 1), lo)
            hi = torch.where(edge, torch.full_like(hi, size - 1), hi)
            v = torch.where(edge, lo.to(f32), v)
            l = v - lo.to(f32)
            return oob, lo, hi, l, 1.0 - l
        oy, yl, yh, ly, hy = prep(ys, H)
        ox, xl, xh, lx, hx = prep(xs, W)
        v1 = feat[:, yl][:, :, xl]
        v2 = feat[:, yl][:, :, xh]
        v3 = feat[:, yh][:, :, xl]
        v4 = feat[:, yh][:, :, xh]
        val = (hy.view(-1, 1) * hx.view(1, -1)) * v1 + (hy.view(-1, 1) * lx.view(1, -1)) * v2 + \
              (ly.view(-1, 1) * hx.view(1, -1)) * v3 + (ly.view(-1, 1) * lx.view(1, -1)) * v4
        val = val * (~oy).view(-1, 1).to(f32) * (~ox).view(1, -1).to(f32)
        val = val.view(C, pooled, gh, pooled, gw).sum((2, 4)) / cnt
        out[r] = val
    return out


def roi_pool_levels(feats: Sequence[Tensor], boxes: Tensor, pooled: int, strides=(8, 16, 32), compiled: bool = False) -> Tensor:
    """ROIPooler.forward for one image (poolers.py:190-250).  feats[l] [1,C,H,W].
    compiled=True: the C twin of roi_align (oracle/ref_decode.c; fp32 eval only) -- what the CPU baseline times."""
    lv = assign_levels(boxes)
    out = torch.zeros(len(boxes), feats[0].shape[1], pooled, pooled)
    for l, f in enumerate(feats):
        idx = torch.nonzero(lv == l).squeeze(1)
        if len(idx):
            if compiled:
                from . import decode as odec
                out[idx] = torch.from_numpy(odec.roi_align_c(f[0].detach().numpy(), boxes[idx].detach().numpy(), 1.0 / strides[l], pooled))
            else:
                out[idx] = roi_align(f[0], boxes[idx], 1.0 / strides[l], pooled)
    return out


def roi_head_features(box_feat: Tensor, support_8: Tensor, sd: SD, prefix: str = "roi_heads.") -> Tensor:
    """_run_stage (fsod_roi_heads.py:459-520), the live branch: DSA mix with the mean support feature -> fc1 -> ReLU.
    box_feat [R,C,8,8], support_8 [N,C,8,8].  (attn_4 / fc2 / fc3 are dead compute in the reference: SURVEY App. C.5.)"""
    s = support_8.mean(0, True).expand_as(box_feat)
    a = dense_conv1x1(torch.cat((box_feat, s), 1), sd[prefix + "conv3.weight"], sd[prefix + "conv3.bias"]) + \
        torch.cat((dense_conv1x1(box_feat, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"]),
                   dense_conv1x1(s, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"])), 1)
    return F.relu(dense_linear(a.flatten(1), sd[prefix + "box_head.0.fc1.weight"], sd[prefix + "box_head.0.fc1.bias"]))


def roi_head_eval(feats: Sequence[Tensor], proposals: Tensor, support_8: Tensor, sd: SD, image_hw: Tuple[int, int],
                  score_thresh: float = 0.0, nms_thresh: float = 0.9, topk: int = 100, prefix: str = "roi_heads.", compiled_roi_align: bool = False):
    """Whole second stage for one image; the predict/NMS part runs in oracle/ref_decode.c (bit-exact twin of the HIP kernel)."""
    from . import decode as odec
    x = roi_pool_levels(feats, proposals, 8, compiled=compiled_roi_align)
    h = roi_head_features(x, support_8, sd, prefix)
    det = odec.roi_predict(h.numpy(), sd[prefix + "box_predictor.0.cls_score.weight"].numpy(),
                           sd[prefix + "box_predictor.0.cls_score.bias"].numpy(),
                           sd[prefix + "box_predictor.0.bbox_pred.weight"].numpy(),
                           sd[prefix + "box_predictor.0.bbox_pred.bias"].numpy(), proposals.numpy(), (10.0, 10.0, 5.0, 5.0),
                           image_hw, score_thresh, nms_thresh, topk)
    det.update(box_features=x, h=h)
    return det


def synth_roi_state(sd: Dict[str, Tensor], seed: int = 0, C: int = 128) -> Dict[str, Tensor]:
    """Adds seeded second-stage parameters (reference shapes) + rcnn_8 support features."""
    g = torch.Generator().manual_seed(seed + 3000)

    def u(*shape, b):
        return (torch.rand(*shape, generator=g) * 2 - 1) * b
    p = "roi_heads."
    sd = dict(sd)
    sd[p + "conv1.weight"], sd[p + "conv1.bias"] = u(C // 2, C, 1, 1, b=C ** -0.5), u(C // 2, b=0.1)
    sd[p + "conv2.weight"], sd[p + "conv2.bias"] = u(C // 2, C, 1, 1, b=C ** -0.5), u(C // 2, b=0.1)
    sd[p + "conv3.weight"], sd[p + "conv3.bias"] = u(C, 2 * C, 1, 1, b=(2 * C) ** -0.5), u(C, b=0.1)
    sd[p + "box_head.0.fc1.weight"], sd[p + "box_head.0.fc1.bias"] = u(128, C * 64, b=(C * 64) ** -0.5 * 3), u(128, b=0.1)
    sd[p + "box_predictor.0.cls_score.weight"], sd[p + "box_predictor.0.cls_score.bias"] = u(2, 128, b=0.5), u(2, b=0.1)
    sd[p + "box_predictor.0.bbox_pred.weight"], sd[p + "box_predictor.0.bbox_pred.bias"] = u(4, 128, b=0.3), u(4, b=0.1)
    sd[p + "fc2.weight"], sd[p + "fc2.bias"] = u(128, C * 16, b=0.02), u(128, b=0.1)
    sd[p + "fc3.weight"], sd[p + "fc3.bias"] = u(128, 256, b=0.06), u(128, b=0.1)
    return sd


# --------------------------------------------------------------------------------------
# a12  CenterNet training targets + losses (ref:fewx/modeling/fsod/fsod_rpn.py:702-779, 803-1065)
# --------------------------------------------------------------------------------------
CN_INF = 100000000


def centernet_grids(shapes: Sequence[Tuple[int, int]], strides: Sequence[int]) -> List[Tensor]:
    """compute_grids (fsod_rpn.py:782-800)."""
    out = []
    for (h, w), s in zip(shapes, strides):
        ys = torch.arange(0, h * s, step=s, dtype=torch.float32)
        xs = torch.arange(0, w * s, step=s, dtype=torch.float32)
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        out.append(torch.stack((xx.reshape(-1), yy.reshape(-1)), 1) + s // 2)
    return out


def centernet_targets(gt_boxes: Sequence[Tensor], shapes: Sequence[Tuple[int, int]], strides=(8, 16, 32),
                      soi=((0, 64), (48, 192), (128, 1000000)), hm_min_overlap=0.8, min_radius=4):
    """_get_ground_truth + _get_label_inds for only_proposal=True.  gt_boxes: per image [N_i,4].
    Returns pos_inds [N'] int64, reg_targets [M*B,4], flattened_hms [M*B] in level-first order."""
    L, B = len(strides), len(gt_boxes)
    delta = (1 - hm_min_overlap) / (1 + hm_min_overlap)
    grids_l = centernet_grids(shapes, strides)
    nloc = [len(g) for g in grids_l]
    strides_m = torch.cat([torch.full((nloc[l],), float(strides[l])) for l in range(L)])
    ranges_m = torch.cat([torch.tensor(soi[l], dtype=torch.float32).view(1, 2).expand(nloc[l], 2) for l in range(L)])
    grids = torch.cat(grids_l, 0)
    M = grids.shape[0]
    regs, hms = [], []
    for boxes in gt_boxes:
        N = boxes.shape[0]
        if N == 0:
            regs.append(torch.zeros(M, 4) - CN_INF)
            hms.append(torch.zeros(M))
            continue
        area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
        l = grids[:, 0].view(M, 1) - boxes[:, 0].view(1, N)
        t = grids[:, 1].view(M, 1) - boxes[:, 1].view(1, N)
        r = boxes[:, 2].view(1, N) - grids[:, 0].view(M, 1)
        b = boxes[:, 3].view(1, N) - grids[:, 1].view(M, 1)
        reg = torch.stack([l, t, r, b], 2)
        centers = (boxes[:, [0, 1]] + boxes[:, [2, 3]]) / 2
        ce = centers.view(1, N, 2).expand(M, N, 2)
        se = strides_m.view(M, 1, 1).expand(M, N, 2)
        cd = ((ce / se).int() * se).float() + se / 2
        ge = grids.view(M, 1, 2).expand(M, N, 2)
        is_peak = ((ge - cd) ** 2).sum(2) == 0
        in_box = reg.min(2)[0] > 0
        c33 = ((ge[:, :, 0] - cd[:, :, 0]).abs() <= se[:, :, 0]) & ((ge[:, :, 1] - cd[:, :, 1]).abs() <= se[:, :, 0]) & in_box
        crit = ((reg[:, :, :2] + reg[:, :, 2:]) ** 2).sum(2) ** 0.5 / 2
        cared = (crit >= ranges_m[:, [0]]) & (crit <= ranges_m[:, [1]])
        mask = c33 & cared
        d2 = ((ge - ce) ** 2).sum(2)
        d2[is_peak] = 0
        r2 = torch.clamp(delta ** 2 * 2 * area, min=min_radius ** 2)
        wd = d2 / r2.view(1, N).expand(M, N)
        dm = wd.clone()
        dm[mask == 0] = CN_INF * 1.0
        mn, mi = dm.min(1)
        rt = reg[range(M), mi]
        rt[mn == CN_INF] = -CN_INF
        hm = torch.exp(-wd.min(1)[0])
        hm[hm < 1e-4] = 0
        regs.append(rt)
        hms.append(hm)
    # level first
    reg_lf = torch.cat([torch.cat([torch.split(x, nloc, 0)[l] for x in regs], 0) / float(strides[l]) for l in range(L)], 0)
    hm_lf = torch.cat([torch.cat([torch.split(x, nloc, 0)[l] for x in hms], 0) for l in range(L)], 0)
    # positive indices
    loc = torch.tensor([h * w for h, w in shapes], dtype=torch.int64)
    bases, s = [], 0
    for l in range(L):
        bases.append(s)
        s += B * int(loc[l])
    bases = torch.tensor(bases, dtype=torch.int64)
    Ws = torch.tensor([w for _, w in shapes], dtype=torch.int64)
    sr = torch.tensor(soi, dtype=torch.float32)
    pos = []
    for im, boxes in enumerate(gt_boxes):
        n = boxes.shape[0]
        if n == 0:
            continue
        c = ((boxes[:, [0, 1]] + boxes[:, [2, 3]]) / 2).view(n, 1, 2).expand(n, L, 2)
        ci = (c / torch.tensor(strides, dtype=torch.float32).view(1, L, 1)).long()
        ind = bases.view(1, L) + im * loc.view(1, L) + ci[:, :, 1] * Ws.view(1, L) + ci[:, :, 0]
        crit = ((boxes[:, 2:] - boxes[:, :2]) ** 2).sum(1) ** 0.5 / 2
        cared = (crit.view(n, 1) >= sr[:, 0].view(1, L)) & (crit.view(n, 1) <= sr[:, 1].view(1, L))
        pos.append(ind[cared].view(-1))
    pos = torch.cat(pos) if pos else torch.zeros(0, dtype=torch.int64)
    return pos, reg_lf, hm_lf


def centernet_losses(reg_pred: Tensor, hm_logit: Tensor, pos_inds: Tensor, reg_targets: Tensor, hms: Tensor, num_gpus: int = 1,
                     reg_weight=1.0, pos_weight=0.5, neg_weight=0.5, alpha=0.25, beta=4, gamma=2.0, clamp=1e-4, ignore_high_fp=0.85):
    """CenterNet.losses for only_proposal + with_agn_hm + not_norm_reg (fsod_rpn.py:702-779); world size 1.
    reg_pred [M,4] (after Scale+ReLU), hm_logit [M].  Returns dict of the three scalar losses + the raw sums."""
    num_pos_avg = max(pos_inds.numel() / num_gpus, 1.0)
    idx = torch.nonzero(reg_targets.max(1)[0] >= 0).squeeze(1)
    p, t = reg_pred[idx], reg_targets[idx]
    ta = (t[:, 0] + t[:, 2]) * (t[:, 1] + t[:, 3])
    pa = (p[:, 0] + p[:, 2]) * (p[:, 1] + p[:, 3])
    wi = torch.min(p[:, 0], t[:, 0]) + torch.min(p[:, 2], t[:, 2])
    hi = torch.min(p[:, 3], t[:, 3]) + torch.min(p[:, 1], t[:, 1])
    gw = torch.max(p[:, 0], t[:, 0]) + torch.max(p[:, 2], t[:, 2])
    gh = torch.max(p[:, 3], t[:, 3]) + torch.max(p[:, 1], t[:, 1])
    ac, ai = gw * gh, wi * hi
    au = ta + pa - ai
    giou = (ai + 1.0) / (au + 1.0) - (ac - au) / ac
    giou_sum = (1 - giou).sum()
    reg_norm = max(float(len(idx)) / num_gpus, 1)
    pred = torch.clamp(torch.sigmoid(hm_logit.float()), min=clamp, max=1 - clamp)
    nw = torch.pow(1 - hms, beta)
    pp = pred[pos_inds]
    pos_sum = (torch.log(pp) * torch.pow(1 - pp, gamma)).sum()
    nl = torch.log(1 - pred) * torch.pow(pred, gamma) * nw
    nl = (pred < ignore_high_fp).float() * nl
    neg_sum = nl.sum()
    return {"loss_centernet_loc": reg_weight * giou_sum / reg_norm,
            "loss_centernet_agn_pos": pos_weight * alpha * (-pos_sum) / num_pos_avg,
            "loss_centernet_agn_neg": neg_weight * (1 - alpha) * (-neg_sum) / num_pos_avg,
            "sums": torch.stack([giou_sum, torch.tensor(float(len(idx))), pos_sum, neg_sum])}


# --------------------------------------------------------------------------------------
# a13  clip + SGD on a flat bucket (torch.nn.utils.clip_grad_value_ + torch.optim.SGD.step; ref:fewx/solver/build.py:18-60,110-139)
# --------------------------------------------------------------------------------------
def sgd_step_flat(params: Tensor, grads: Tensor, momentum_buf: Tensor, chunk_lr: Tensor, chunk_wd: Tensor, lr_scale: float,
                  momentum: float, clip_value: float, grad_scale: float) -> None:
    """In-place; chunk_* are per 256 elements."""
    lr = (chunk_lr * lr_scale).repeat_interleave(256)
    wd = chunk_wd.repeat_interleave(256)
    g = grads * grad_scale
    if clip_value > 0:
        g = g.clamp(-clip_value, clip_value)
    g = g + wd * params
    momentum_buf.mul_(momentum).add_(g)
    params.sub_(lr * momentum_buf)
