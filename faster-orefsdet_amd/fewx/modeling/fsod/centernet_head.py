"""CenterNet2 proposal head (ref:CenterNet2/centernet/modeling/dense_heads/centernet_head.py:13-161) in the one
configuration the path uses: only_proposal, with_agn_hm, NUM_BOX_CONVS=1, NORM=GN (log:697-715).

HIP schedule per level: tower conv3x3+bias -> GroupNorm statistics folded to a per-channel affine ->
ONE conv3x3 producing (l,t,r,b | hm) with the GN affine + ReLU applied to its input on the fly and
Scale_l / bias / ReLU(reg only) in its epilogue.  Parameter names: bbox_tower.0, bbox_tower.1 (GN), bbox_pred,
agn_hm, scales.{l}.scale."""
import math
from typing import List

import torch
from torch import nn

from detectron2.layers import nhwc_view, _require_gpu


class Scale(nn.Module):
    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.FloatTensor([init_value]))

    def forward(self, x):
        return x * self.scale


class CenterNetHead(nn.Module):
    def __init__(self, in_channels, num_levels, *, num_classes=80, with_agn_hm=False, only_proposal=False, norm="GN",
                 num_cls_convs=4, num_box_convs=4, num_share_convs=0, use_deformable=False, prior_prob=0.01):
        super().__init__()
        if not (only_proposal and with_agn_hm and norm == "GN" and num_share_convs == 0 and not use_deformable
                and num_box_convs == 1 and in_channels % 32 == 0):
            raise NotImplementedError("CenterNetHead: only ONLY_PROPOSAL + WITH_AGN_HM + GN + NUM_BOX_CONVS=1 is built "
                                      "(the finetune_vovnet.yaml configuration)")
        self.num_classes, self.with_agn_hm, self.only_proposal = num_classes, True, True
        self.cls_tower = nn.Sequential()
        self.share_tower = nn.Sequential()
        self.bbox_tower = nn.Sequential(nn.Conv2d(in_channels, in_channels, 3, 1, 1, bias=True), nn.GroupNorm(32, in_channels), nn.ReLU())
        self.bbox_pred = nn.Conv2d(in_channels, 4, 3, 1, 1)
        self.scales = nn.ModuleList([Scale(1.0) for _ in range(num_levels)])
        self.agn_hm = nn.Conv2d(in_channels, 1, 3, 1, 1)
        for l in [self.bbox_tower[0], self.bbox_pred, self.agn_hm]:
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)
        nn.init.constant_(self.bbox_pred.bias, 8.0)
        nn.init.constant_(self.agn_hm.bias, -math.log((1 - prior_prob) / prior_prob))
        self._cache = None

    @classmethod
    def from_config(cls, cfg, input_shape):
        c = cfg.MODEL.CENTERNET
        return dict(in_channels=[s.channels for s in input_shape][0], num_levels=len(input_shape), num_classes=c.NUM_CLASSES,
                    with_agn_hm=c.WITH_AGN_HM, only_proposal=c.ONLY_PROPOSAL, norm=c.NORM, num_cls_convs=c.NUM_CLS_CONVS,
                    num_box_convs=c.NUM_BOX_CONVS, num_share_convs=c.NUM_SHARE_CONVS, use_deformable=c.USE_DEFORMABLE,
                    prior_prob=c.PRIOR_PROB)

    def _packed(self):
        import orehip
        ps = [self.bbox_tower[0].weight, self.bbox_pred.weight, self.agn_hm.weight]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._cache is None or self._cache[0] != key:
            w5 = torch.cat([self.bbox_pred.weight.detach(), self.agn_hm.weight.detach()], 0)
            self._cache = (key, orehip.pack_conv_weight(self.bbox_tower[0].weight), orehip.pack_conv_weight(w5))
        return self._cache[1], self._cache[2]

    def forward_nhwc(self, feats: List[torch.Tensor]) -> List[torch.Tensor]:
        """feats[l] [B,H,W,C] -> head[l] [B,H,W,8]: channels 0..3 = relu(scale_l*(reg)), 4 = agn heatmap logit."""
        import orehip
        wt, w5 = self._packed()
        C = self.bbox_pred.in_channels
        gn = self.bbox_tower[1]
        outs = []
        for l, x in enumerate(feats):
            t = orehip.conv2d(x, wt, C, 3, shift=self.bbox_tower[0].bias.detach())
            mul, add = orehip.groupnorm_affine(t, gn.num_groups, gn.weight.detach(), gn.bias.detach(), gn.eps)
            s = self.scales[l].scale.detach()
            scale = torch.cat([s.expand(4), torch.ones(1, device=s.device)])
            shift = torch.cat([self.bbox_pred.bias.detach() * s, self.agn_hm.bias.detach()])
            out = torch.zeros(*x.shape[:3], 8, device=x.device, dtype=torch.float32)
            orehip.conv2d(t, w5, 5, 3, scale=scale.contiguous(), shift=shift.contiguous(), relu_cout=4, in_mul=mul, in_add=add,
                          in_relu=True, out=out)
            outs.append(out)
        return outs

    def forward(self, x: List[torch.Tensor]):
        """Reference protocol: returns (clss, bbox_reg, agn_hms) per level as NCHW tensors (clss entries are None)."""
        for f in x:
            _require_gpu(f, "CenterNetHead")
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or any(f.requires_grad for f in x)):
            from .train_forward import head_train            # autograd bindings: conv fwd/dgrad/wgrad, GroupNorm fwd/bwd kernels
            heads = head_train(self, [nhwc_view(f) for f in x])
        else:
            heads = self.forward_nhwc([nhwc_view(f) for f in x])
        regs = [h[..., :4].permute(0, 3, 1, 2) for h in heads]
        hms = [h[..., 4:5].permute(0, 3, 1, 2) for h in heads]
        return [None] * len(heads), regs, hms
