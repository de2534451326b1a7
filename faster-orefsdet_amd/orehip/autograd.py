"""torch.autograd bindings of the HIP forward/backward kernels.

torch.autograd is used as the TAPE only: it records which kernels ran and replays their backward entry points in reverse
order (what `losses.backward()` does in d2z:engine/train_loop.py:279).  Every FLOP-carrying node is a libore_hip.so call:
    ConvFn      conv/linear (+ FrozenBN scale/shift or bias, + ReLU): ore_conv2d_fwd | ore_relu_affine_bwd, ore_conv2d_fwd with
                the data-gradient weight layout, ore_conv2d_wgrad_fwd, ore_colsum_fwd
    OSAFn       one whole OSA block over its concat buffer (d2z:modeling/backbone/vovnet.py:310-332): layers write channel
                slices, the backward walks the slices of ONE gradient buffer (no torch.cat / split)
    RoiAlignFn  ore_roi_align_fwd | ore_roi_align_bwd
    CenterNetLossFn  ore_centernet_losses_fwd | ore_centernet_losses_bwd (normalisers stay on device)
Weights are repacked on device (ore_pack_conv_weight_fwd) at most once per optimizer step and layout.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch.autograd import Function

import orehip

_EPOCH = [0]


def weights_changed() -> None:
    """Invalidate every packed copy (called where weights are rewritten behind torch's back, and inside a captured training graph so
    that each replay repacks the current parameters)."""
    _EPOCH[0] += 1


def weights_epoch() -> int:
    return _EPOCH[0]


def _base_param(w: torch.Tensor):
    b = w._base if w._base is not None else w
    return b if isinstance(b, torch.nn.Parameter) else None


def packed(w: torch.Tensor, dgrad: bool) -> torch.Tensor:
    """MFMA-packed copy of a conv / linear weight.  Only parameters and views of parameters are cached, and the cache lives ON the
    parameter object (it dies with the model; an address- or id-keyed table can hand a new model the packed weights of a freed one):
    key = the view's geometry, tag = the weight epoch, the parameter's version counter (views share it) and its storage address.
    A temporary built from parameters (e.g. the concatenated (reg|hm) head weight) is packed afresh on every call."""
    base = _base_param(w)
    if base is None:
        return orehip.pack_conv_weight_dev(w.detach().contiguous(), dgrad)
    cache = base.__dict__.setdefault("_ore_packed", {})
    _PACKED_OWNERS[id(base)] = base
    key = (tuple(w.shape), tuple(w.stride()), w.storage_offset(), bool(dgrad))
    tag = (_EPOCH[0], base._version, base.data_ptr())
    e = cache.get(key)
    if e is None or e[0] != tag:
        cache[key] = e = (tag, orehip.pack_conv_weight_dev(w.detach().contiguous(), dgrad, out=e[1] if e is not None else None))
    return e[1]


_PACKED_OWNERS = __import__("weakref").WeakValueDictionary()   # id -> parameter that carries a packed-weight cache (tensors do not hash by value)


def prepack(params=None) -> int:
    """Refresh every STALE cached packed weight (those packed() has been asked for before: both layouts of every conv / linear of the
    previous step) in one launch -- at the top of a training forward, instead of one 4 us kernel per weight and layout scattered
    over the step (86 per step).  Returns the number of weights repacked.  Views that are not contiguous are left to packed()."""
    jobs, fresh = [], []
    owners = list(_PACKED_OWNERS.values()) if params is None else [p for p in params if "_ore_packed" in p.__dict__]
    for base in owners:
        cache = base.__dict__.get("_ore_packed")
        if not cache:
            continue
        tag = (_EPOCH[0], base._version, base.data_ptr())
        for key, (etag, out) in cache.items():
            if etag == tag:
                continue
            shape, stride, off, dgrad = key
            w = base.detach().as_strided(shape, stride, off)
            if not w.is_contiguous() or len(shape) != 4:
                continue
            jobs.append((w, dgrad, out))
            fresh.append((cache, key, tag, out))
    for i in range(0, len(jobs), 256):
        orehip.pack_conv_weights_multi(jobs[i:i + 256])
    for cache, key, tag, out in fresh:
        cache[key] = (tag, out)
    return len(jobs)


def _c16(n: int) -> int:
    return (n + 15) // 16 * 16


def packed_wino(w: torch.Tensor, dgrad: bool):
    """Winograd F(2x2,3x3) form of packed(w, dgrad) for the 3x3 layers the Winograd kernels cover (ore_winograd_covers), else None.
    Cached next to the packed copy, same tag."""
    if w.dim() != 4 or w.shape[2] != 3 or w.shape[3] != 3:
        return None
    cout, cin = (w.shape[1], _c16(w.shape[0])) if dgrad else (w.shape[0], w.shape[1])     # of the conv that will be launched
    if not orehip.winograd_covers(cout, cin):
        return None
    pw = packed(w, dgrad)
    base = _base_param(w)
    if base is None:
        return orehip.winograd_weight(pw, cout, cin)
    cache = base.__dict__.setdefault("_ore_wino", {})
    key = (tuple(w.shape), tuple(w.stride()), w.storage_offset(), bool(dgrad))
    tag = (_EPOCH[0], base._version, base.data_ptr())
    e = cache.get(key)
    if e is None or e[0] != tag:
        cache[key] = e = (tag, orehip.winograd_weight(pw, cout, cin))
    return e[1]


DIRECT_GRAD_OFF = False      # set while a backward is run for its RESULT only (torch.autograd.grad inside make_graphed_callables: its warm-up
                             # passes must not leave gradients in the bucket)


def direct_grad(p) -> Optional[torch.Tensor]:
    """The buffer a parameter's gradient may be accumulated into BY THE KERNEL (dW = 1 * grad + ...; the backward then hands autograd no
    gradient for it): its `.grad`, when the owner of that buffer allows it (`p._ore_direct_grad`, set by fewx.solver.FlatBucket whose
    zero_grad() zeroes the buffer every step) -- else None.  The backbone runs twice per step (query and support branch): without this
    every trainable conv weight costs an add of the two branch gradients plus an in-place add into `.grad`, ~100 launches of a few
    microseconds per step.  Not under a data-parallel wrapper (its exchange is issued from post-accumulate hooks, which need the
    engine's AccumulateGrad to run) and not with tensor hooks on the parameter."""
    if p is None or DIRECT_GRAD_OFF or not getattr(p, "_ore_direct_grad", False):
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape:
        return None
    return g


def _conv_backward(x, x_coff, Cin, weight, dz, k, need_x: bool, need_w: bool, need_b: bool, bias=None):
    """dz [B,H,W,Cout16] contiguous -> (dX [B,H,W,Cin] | None, dW | None, db | None).  A gradient accumulated in place (direct_grad)
    comes back as None."""
    Cout = weight.shape[0]
    gx = gw = gb = None
    if need_x:
        gx = orehip.conv2d(dz, packed(weight, True), Cin, k, 1, k // 2, w_wino=packed_wino(weight, True) if k == 3 else None)
    full = dz.shape[-1] == Cout                              # (a Cout padded to 16 has no in-place form: the kernel writes Cout16 rows)
    dgw = direct_grad(weight) if need_w and full else None
    dgb = direct_grad(bias) if need_b and full else None
    if dgw is not None and (not need_b or dgb is not None):
        orehip.conv2d_wgrad(x, dz, k, x_coff=x_coff, Cin=Cin, out=dgw.view(Cout, Cin, k, k), beta=1.0, want_bias=need_b, db_out=dgb, beta_b=1.0)
        return gx, None, None
    if need_w and need_b:                                    # the weight-gradient launch sums the dZ rows it stages anyway
        gw, gb = orehip.conv2d_wgrad(x, dz, k, x_coff=x_coff, Cin=Cin, want_bias=True)
        gw, gb = gw[:Cout], gb[:Cout]
    elif need_w:
        gw = orehip.conv2d_wgrad(x, dz, k, x_coff=x_coff, Cin=Cin)[:Cout]
    elif need_b:
        gb = orehip.colsum(dz)[:Cout]
    return gx, gw, gb


class ConvFn(Function):
    """y = act(conv(x, W) * scale + shift) or act(conv(x, W) + bias);  x [B,H,W,Cin] NHWC, W OIHW, stride 1, pad k//2."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale, shift, relu: bool, add=None):
        """add: optional [B,ceil(H/2),ceil(W/2),Cout] summed in nearest-2x upsampled (the FPN top-down path) in the conv epilogue."""
        x = x.contiguous()
        Cout, Cin, k, _ = weight.shape
        co16 = _c16(Cout)
        out = torch.empty(*x.shape[:3], co16, device=x.device, dtype=torch.float32) if co16 != Cout else None
        sh = shift if shift is not None else (bias.detach().contiguous() if bias is not None else None)
        if out is not None:
            out.zero_()
        assert add is None or (co16 == Cout and not relu)
        y = orehip.conv2d(x, packed(weight, False), Cout, k, 1, k // 2, scale=scale, shift=sh, relu_cout=Cout if relu else 0, out=out,
                          add=add.contiguous() if add is not None else None, w_wino=packed_wino(weight, False) if k == 3 and add is None else None)
        ctx.save_for_backward(x, weight, scale, y if relu else None)
        ctx.bias_ref = bias                                     # (only looked at for direct_grad; its values are not needed)
        ctx.meta = (k, relu, bias is not None, Cout, Cin, co16)
        ctx.has_add = add is not None
        return y if co16 == Cout else y[..., :Cout]

    @staticmethod
    def backward(ctx, dy):
        x, weight, scale, y = ctx.saved_tensors
        k, relu, has_bias, Cout, Cin, co16 = ctx.meta
        if co16 != Cout:
            dz = torch.zeros(*dy.shape[:3], co16, device=dy.device, dtype=torch.float32)
            dz[..., :Cout] = dy
        else:
            dz = dy.contiguous()
        if relu:
            dz = orehip.relu_affine_bwd(dz, y, scale)
        elif scale is not None:
            dz = dz * scale
        gx, gw, gb = _conv_backward(x, 0, Cin, weight, dz, k, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                    has_bias and ctx.needs_input_grad[2], bias=ctx.bias_ref)
        gadd = orehip.sumpool2x2(dz) if (ctx.has_add and ctx.needs_input_grad[6]) else None
        return gx, gw, gb, None, None, None, gadd


def conv(x, weight, bias=None, scale=None, shift=None, relu=False, add=None):
    return ConvFn.apply(x, weight, bias, scale, shift, relu, add)


def linear(x2d, weight, bias=None, relu=False):
    """F.linear (+ReLU) on rows: x2d [N, in] -> [N, out]."""
    N, cin = x2d.shape
    y = ConvFn.apply(x2d.reshape(1, 1, N, cin), weight.reshape(weight.shape[0], cin, 1, 1), bias, None, None, relu, None)
    return y.reshape(N, weight.shape[0])


class OSAFn(Function):
    """One OSA block up to (not including) eSE: n x [conv3x3 + FrozenBN + ReLU] chained through channel slices of one concat
    buffer, then conv1x1 + FrozenBN + ReLU over the whole buffer.  args: x_in, then per layer (weight, scale, shift), then the
    concat conv's (weight, scale, shift).  d2z:modeling/backbone/vovnet.py:310-332."""

    @staticmethod
    def forward(ctx, x_in, *args):
        n = len(args) // 3 - 1
        ws, scs, shs = args[0::3], args[1::3], args[2::3]
        x_in = x_in.contiguous()
        B, H, W, in_ch = x_in.shape
        stage_ch = ws[0].shape[0]
        cat_ch = in_ch + n * stage_ch
        cat = torch.empty(B, H, W, cat_ch, device=x_in.device, dtype=torch.float32)
        cat[..., :in_ch] = x_in
        src, cin, dst = 0, in_ch, in_ch
        for i in range(n):
            orehip.conv2d(cat, packed(ws[i], False), stage_ch, 3, 1, 1, in_coff=src, Cin=cin, scale=scs[i], shift=shs[i],
                          relu_cout=stage_ch, out=cat, out_coff=dst, w_wino=packed_wino(ws[i], False))
            src, cin, dst = dst, stage_ch, dst + stage_ch
        y = orehip.conv2d(cat, packed(ws[n], False), ws[n].shape[0], 1, 1, 0, scale=scs[n], shift=shs[n], relu_cout=ws[n].shape[0])
        ctx.save_for_backward(cat, y, *ws, *scs)
        ctx.meta = (n, in_ch, stage_ch)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, in_ch, stage_ch = ctx.meta
        saved = ctx.saved_tensors
        cat, y = saved[0], saved[1]
        ws, scs = saved[2:2 + n + 1], saved[3 + n:3 + n + n + 1]
        grads: List[Optional[torch.Tensor]] = [None] * (3 * (n + 1))
        dz = orehip.relu_affine_bwd(dy.contiguous(), y, scs[n])
        def wgrad(i, xbuf, dzi, k, **kw):                       # into the parameter's gradient where allowed, else returned to autograd
            dg = direct_grad(ws[i])
            if dg is not None and dzi.shape[-1] == ws[i].shape[0]:
                orehip.conv2d_wgrad(xbuf, dzi, k, out=dg, beta=1.0, **kw)
                return None
            return orehip.conv2d_wgrad(xbuf, dzi, k, **kw)

        if ctx.needs_input_grad[1 + 3 * n]:
            grads[3 * n] = wgrad(n, cat, dz, 1)
        dcat = orehip.conv2d(dz, packed(ws[n], True), cat.shape[-1], 1, 1, 0)          # gradient of every concat slice
        for i in range(n - 1, -1, -1):
            dst = in_ch + i * stage_ch
            src, cin = (0, in_ch) if i == 0 else (dst - stage_ch, stage_ch)
            dzi = orehip.relu_affine_bwd(dcat, cat, scs[i], dy_coff=dst, y_coff=dst, Cc=stage_ch)
            if ctx.needs_input_grad[1 + 3 * i]:
                grads[3 * i] = wgrad(i, cat, dzi, 3, x_coff=src, Cin=cin)
            if i > 0 or ctx.needs_input_grad[0]:
                # (.add_ on the view: `dcat[...] += x` would follow the in-place add with a copy of the slice onto itself)
                dcat[..., src:src + cin].add_(orehip.conv2d(dzi, packed(ws[i], True), cin, 3, 1, 1, w_wino=packed_wino(ws[i], True)))
        gx = dcat[..., :in_ch].contiguous() if ctx.needs_input_grad[0] else None
        return (gx, *grads)


def osa_block(x_in, layers: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]):
    flat = [t for tri in layers for t in tri]
    return OSAFn.apply(x_in, *flat)


_UNIT = {}


def _unit_affine(C: int, device):
    """(ones[C], zeros[C]) on `device`, created once (GroupNorm statistics are taken with a unit affine)."""
    key = (C, str(device))
    if key not in _UNIT:
        if torch.cuda.is_current_stream_capturing():          # a fill captured into a graph only runs on replay: do not cache it
            return torch.ones(C, device=device), torch.zeros(C, device=device)
        _UNIT[key] = (torch.ones(C, device=device), torch.zeros(C, device=device))
    return _UNIT[key]


class GroupNormReluFn(Function):
    """relu?(GroupNorm(x)) for x [B,H,W,C], statistics per image (head tower: GN(32,128) + ReLU).  Statistics by the engine's
    Chan-combine kernels (ore_groupnorm_affine_fwd with unit gamma), apply + backward in ore_groupnorm_apply_fwd / ore_groupnorm_bwd;
    the B images of a batch go through each kernel in one launch."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups: int, eps: float, relu: bool):
        x = x.contiguous()
        C = x.shape[-1]
        one, zero = _unit_affine(C, x.device)
        r, a = orehip.groupnorm_affine(x, groups, one, zero, eps)              # [B,C]: rstd, -mean*rstd of the channel's group
        y = orehip.groupnorm_apply(x, r, a, gamma.detach().contiguous(), beta.detach().contiguous(), relu)
        ctx.save_for_backward(x, y, r, a, gamma)
        ctx.meta = (groups, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, r, a, gamma = ctx.saved_tensors
        groups, relu = ctx.meta
        dx, dbeta, dgamma = orehip.groupnorm_bwd(dy.contiguous(), y, x, groups, r, a, gamma.detach().contiguous(), relu)
        if dbeta.shape[0] == 1:
            return dx, dgamma[0], dbeta[0], None, None, None
        return dx, dgamma.sum(0), dbeta.sum(0), None, None, None


def group_norm_relu(x, gamma, beta, groups, eps=1e-5, relu=True):
    return GroupNormReluFn.apply(x, gamma, beta, groups, eps, relu)


class EseFn(Function):
    """eSE (d2z:modeling/backbone/vovnet.py:238-260): y = x * hsigmoid(fc(mean_hw(x))).  x [B,H,W,C]; fc_w [C,C,1,1]; fc_b [C].
    Pixel-sized work (average pool, x*gate, sum_hw dy*x, dy*gate + const) runs in HIP kernels; the [B,C]-sized vector algebra of
    the gate (a C x C mat-vec and its transpose) stays on torch tensors."""

    @staticmethod
    def forward(ctx, x, fc_w, fc_b):
        x = x.contiguous()
        B, H, W, C = x.shape
        m = orehip.prod_colsum(x, None, 1.0 / (H * W))                                 # [B,C] average pool
        z = torch.addmm(fc_b, m, fc_w.reshape(C, C).t())
        g = F.hardsigmoid(z)                                                           # relu6(z + 3) / 6 in one launch
        y = orehip.scale_add_channels(x, g.contiguous(), None)
        ctx.save_for_backward(x, fc_w, m, z, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, fc_w, m, z, g = ctx.saved_tensors
        B, H, W, C = x.shape
        dy = dy.contiguous()
        dg = orehip.prod_colsum(dy, x, 1.0)                                            # [B,C] = sum_hw dy * x
        dz = torch.ops.aten.hardsigmoid_backward(dg, z)                                # dg / 6 where -3 < z < 3, else 0 (one launch)
        Wm = fc_w.reshape(C, C)
        dm = dz @ Wm                                                                   # [B,C]
        dx = orehip.scale_add_channels(dy, g.contiguous(), (dm / (H * W)).contiguous())
        return dx, (dz.t() @ m).reshape(fc_w.shape), dz.sum(0)


def ese(x, fc_w, fc_b):
    return EseFn.apply(x, fc_w, fc_b)


class AdaptiveAvgPoolFn(Function):
    """F.adaptive_avg_pool2d on NHWC maps (ref fsod_cen.py:214-231): x [B,H,W,C] -> [B,OH,OW,C]; deterministic gather backward."""

    @staticmethod
    def forward(ctx, x, OH: int, OW: int):
        x = x.contiguous()
        ctx.meta = (x.shape[1], x.shape[2], OH, OW)
        return orehip.adaptive_avgpool_nhwc(x, OH, OW)

    @staticmethod
    def backward(ctx, dy):
        H, W, OH, OW = ctx.meta
        return orehip.adaptive_avgpool_nhwc(dy.contiguous(), OH, OW, grad_of=(H, W)), None, None


def adaptive_avg_pool(x_nhwc, OH, OW):
    return AdaptiveAvgPoolFn.apply(x_nhwc, OH, OW)


class GroupMeanFn(Function):
    """[groups*N, ...] -> [groups, ...]: mean over each group's N consecutive members (the support prototype over an image's shots)."""

    @staticmethod
    def forward(ctx, x, groups: int):
        x = x.contiguous()
        ctx.meta = (groups, x.shape[0] // groups)
        return orehip.group_mean(x, groups)

    @staticmethod
    def backward(ctx, dy):
        groups, N = ctx.meta
        return orehip.group_mean(dy.contiguous(), groups, backward=True, members=N), None


def group_mean(x, groups):
    return GroupMeanFn.apply(x, groups)


class SmPermuteFn(Function):
    """SM_Block mixing layouts (ref fsod_cen.py:602-611): the permute + reshape pairs around mlp_h / mlp_w as one coalesced granule
    transpose each way; the backward of a layout change is the opposite layout change."""

    @staticmethod
    def forward(ctx, x, dims, axis: str, inverse: bool):
        ctx.meta = (dims, axis, inverse, tuple(x.shape))
        return orehip.sm_permute(x.contiguous(), *dims, axis, inverse)

    @staticmethod
    def backward(ctx, dy):
        dims, axis, inverse, shape = ctx.meta
        return orehip.sm_permute(dy.contiguous(), *dims, axis, not inverse).reshape(shape), None, None, None


def sm_permute(x, dims, axis, inverse=False):
    return SmPermuteFn.apply(x, dims, axis, inverse)


class SmDualPermuteFn(Function):
    """Both mixing layouts of one map, (x in the 'h' layout, x in the 'w' layout); backward: the two gradients come back through the
    inverse layouts into ONE tensor (the second transpose accumulates), instead of two tensors and an add."""

    @staticmethod
    def forward(ctx, x, dims):
        x = x.contiguous()
        ctx.meta = (dims, tuple(x.shape))
        return orehip.sm_permute(x, *dims, "h", False), orehip.sm_permute(x, *dims, "w", False)

    @staticmethod
    def backward(ctx, dh, dw):
        dims, shape = ctx.meta
        dx = orehip.sm_permute(dh.contiguous(), *dims, "h", True)
        orehip.sm_permute(dw.contiguous(), *dims, "w", True, out=dx, accumulate=True)
        return dx.reshape(shape), None


def sm_dual_permute(x, dims):
    return SmDualPermuteFn.apply(x, dims)


class SMTailFn(Function):
    """The tail of the SM_Block (ref fsod_cen.py:612-615) as one node: m = mean_hw(h + w); a = softmax(reweighting(m)); y = w a0 + h a1.
    The [B, C]-sized MLP + softmax stay torch ops, differentiated by an inner autograd graph kept in the node; the map-sized work is
    three HIP passes forward (two pooled sums, the re-weighted sum) and three backward (two per-image column sums of dy * map, and ONE
    pass that writes dw = dy a0 + dm / HW and dh = dy a1 + dm / HW -- the mean's gradient rides along instead of two broadcast adds).
    NOT used by the model: the inner torch.autograd.grad makes the backward re-entrant, and the re-entrant engine stalls the launch
    thread (kernel time of a bs16 step 45.1 -> 44.7 ms, wall time 44.4 -> 49-53 ms; profiles/EXPERIMENTS.md).  Kept as a tested
    reference of the fused arithmetic (tests/test_hip_train.py)."""

    @staticmethod
    def forward(ctx, w, h, mod, *params):
        w, h = w.contiguous(), h.contiguous()
        B, Cc = w.shape[0], w.shape[-1]
        n = w.numel() // (B * Cc)
        m = orehip.prod_colsum(h, None, 1.0 / n) + orehip.prod_colsum(w, None, 1.0 / n)
        with torch.enable_grad():
            m_ = m.detach().requires_grad_(True)
            a = mod(m_).reshape(B, Cc, 2).permute(2, 0, 1).softmax(0)             # [2, B, C]
        a0, a1 = a[0].detach().contiguous(), a[1].detach().contiguous()
        ctx.save_for_backward(w, h, a0, a1)
        ctx.inner = (m_, a, params)
        ctx.n = n
        return orehip.combine2(w, h, a0, a1)

    @staticmethod
    def backward(ctx, dy):
        w, h, a0, a1 = ctx.saved_tensors
        m_, a, params = ctx.inner
        dy = dy.contiguous()
        da = torch.stack([orehip.prod_colsum(dy, w, 1.0), orehip.prod_colsum(dy, h, 1.0)])
        need = [p for p in params if p.requires_grad]
        grads = torch.autograd.grad([a], [m_] + need, [da], allow_unused=True)
        dw, dh = orehip.combine2_bwd(dy, a0, a1, (grads[0] / ctx.n).contiguous())
        it = iter(grads[1:])
        return (dw, dh, None) + tuple(next(it) if p.requires_grad else None for p in params)


def sm_tail(w, h, mod):
    return SMTailFn.apply(w, h, mod, *tuple(mod.parameters()))


class MeanPairFn(Function):
    """m[b, c] = mean over pixels of (h + w) without the sum tensor (ref fsod_cen.py:612); backward: the same constant for every pixel
    of both maps (returned as an expanded view: autograd adds it to the other gradient of h / w in one pass)."""

    @staticmethod
    def forward(ctx, h, w):
        B, Cc = h.shape[0], h.shape[-1]
        n = h.numel() // (B * Cc)
        ctx.meta = (tuple(h.shape), n)
        return orehip.prod_colsum(h.contiguous(), None, 1.0 / n) + orehip.prod_colsum(w.contiguous(), None, 1.0 / n)

    @staticmethod
    def backward(ctx, dm):
        shape, n = ctx.meta
        g = (dm / n).reshape(shape[0], *([1] * (len(shape) - 2)), shape[-1]).expand(shape)
        return g, g


def mean_pair(h, w):
    return MeanPairFn.apply(h, w)


class Combine2Fn(Function):
    """y = w * a0 + h * a1 with per-(image, channel) weights a0, a1 [B, C] (ref fsod_cen.py:614-615) in one pass; backward: both map
    gradients from one pass over dy, the weight gradients as per-image column sums of dy * w and dy * h."""

    @staticmethod
    def forward(ctx, w, h, a0, a1):
        w, h, a0, a1 = w.contiguous(), h.contiguous(), a0.contiguous(), a1.contiguous()
        ctx.save_for_backward(w, h, a0, a1)
        return orehip.combine2(w, h, a0, a1)

    @staticmethod
    def backward(ctx, dy):
        w, h, a0, a1 = ctx.saved_tensors
        dy = dy.contiguous()
        dw, dh = orehip.combine2_bwd(dy, a0, a1)
        return dw, dh, orehip.prod_colsum(dy, w, 1.0), orehip.prod_colsum(dy, h, 1.0)


def combine2(w, h, a0, a1):
    return Combine2Fn.apply(w, h, a0, a1)


class MaxPoolFn(Function):
    """MaxPool2d(3, 2, ceil_mode=True) on NHWC: ore_maxpool3x3s2_fwd | ore_maxpool3x3s2_bwd."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return orehip.maxpool3x3s2(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return orehip.maxpool3x3s2_bwd(x, dy.contiguous())


def maxpool(x):
    return MaxPoolFn.apply(x)


class CorrelationFn(Function):
    """Depthwise support correlation (fsod_cen.py:229-245): q [B,H,W,C]; k11 [C], k13 [C,3], k31 [C,3] (shared) or [B,C], [B,C,3],
    [B,C,3] (every image its own support kernels: a training batch in ONE launch per kernel) -> [B,H,W,2C] = [attn | q], the input
    of conv3 (no torch.cat)."""

    @staticmethod
    def forward(ctx, q, k11, k13, k31):
        q, k11, k13, k31 = q.contiguous(), k11.contiguous(), k13.contiguous(), k31.contiguous()
        cat, t, u = orehip.correlation_train_fwd(q, k11, k13, k31)
        ctx.save_for_backward(q, k11, k13, k31, t, u)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        q, k11, k13, k31, t, u = ctx.saved_tensors
        return orehip.correlation_train_bwd(q, k11, k13, k31, dcat.contiguous(), t, u)


def correlation_cat(q, k11, k13, k31):
    return CorrelationFn.apply(q, k11, k13, k31)


class RoiAlignFn(Function):
    """ROIPooler(8x8, ROIAlignV2, sampling_ratio 0) of ONE image's pyramid: feats[l] [H,W,C] NHWC -> [n, 64, C]."""

    @staticmethod
    def forward(ctx, boxes, strides, pooled, *feats):
        feats = [f.contiguous() for f in feats]
        out = orehip.roi_align(feats, boxes, strides=strides, pooled=pooled)
        ctx.save_for_backward(boxes, *feats)
        ctx.meta = (tuple(strides), pooled)
        return out[:boxes.shape[0]]

    @staticmethod
    def backward(ctx, dout):
        boxes, feats = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        strides, pooled = ctx.meta
        d = orehip.roi_align_bwd(dout.contiguous(), feats, boxes, strides=strides, pooled=pooled)
        return (None, None, None, *d)


def roi_align(feats: Sequence[torch.Tensor], boxes: torch.Tensor, strides=(8, 16, 32), pooled: int = 8) -> torch.Tensor:
    return RoiAlignFn.apply(boxes.detach().float().contiguous(), tuple(strides), pooled, *feats)


class RoiAlignBatchedFn(Function):
    """One box per image of a batch (the support crops): feats[l] [B,H,W,C], boxes [n,4], box_image [n] int32 -> [n, P*P*C]."""

    @staticmethod
    def forward(ctx, boxes, box_image, strides, pooled, *feats):
        feats = [f.contiguous() for f in feats]
        out = orehip.roi_align_batched(feats, boxes, box_image, strides=strides, pooled=pooled)
        ctx.save_for_backward(boxes, box_image, *feats)
        ctx.meta = (tuple(strides), pooled)
        return out

    @staticmethod
    def backward(ctx, dout):
        boxes, box_image, feats = ctx.saved_tensors[0], ctx.saved_tensors[1], ctx.saved_tensors[2:]
        strides, pooled = ctx.meta
        d = orehip.roi_align_bwd(dout.contiguous(), feats, boxes, strides=strides, pooled=pooled, box_image=box_image)
        return (None, None, None, None, *d)


def roi_align_batched(feats, boxes, box_image, strides=(8, 16, 32), pooled: int = 8):
    return RoiAlignBatchedFn.apply(boxes.detach().float().contiguous(), box_image, tuple(strides), pooled, *feats)


class RoiLossFn(Function):
    """loss_cls_stage0 / loss_box_reg_stage0 of the sampled ROIs of B images (custom_fast_rcnn.py:52-81) with their gradients from the
    same launch (ore_roi_losses_fwd); backward scales the stored gradients by the upstream ones."""

    @staticmethod
    def forward(ctx, scores, deltas, boxes, gt, labels, valid, B: int, R: int, reg_weights):
        out, ds, dd = orehip.roi_losses(scores.contiguous(), deltas.contiguous(), boxes.contiguous(), gt.contiguous(), labels, valid, B, R,
                                        reg_weights)
        ctx.save_for_backward(ds, dd)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_cls, g_box):
        ds, dd = ctx.saved_tensors
        return ds * g_cls, dd * g_box, None, None, None, None, None, None, None


def roi_losses(scores, deltas, boxes, gt, labels, valid, B, R, reg_weights):
    return RoiLossFn.apply(scores, deltas, boxes, gt, labels, valid, B, R, tuple(float(v) for v in reg_weights))


class CenterNetLossFn(Function):
    """head [rows, ld>=5] (cols 0..3 ltrb after Scale+ReLU, col 4 heatmap logit) -> [loss_loc, loss_agn_pos, loss_agn_neg, reg rows,
    positives] (ref:fewx/modeling/fsod/fsod_rpn.py:702-779; the last two are this rank's un-normalised counts, no gradient).
    Normalisers exactly the reference's (fsod_rpn.py:712-716,748-751): `max(reduce_sum(n) / num_gpus, 1)` -- this rank's sums over ALL
    its images divided by the per-rank average of the totals, clamped at 1.  (Rounds 2-3 clamped at the number of images on the rank,
    `max(total / world, images)`, so that B images on one rank and B single-image ranks agreed even in the clamped regime; that was this
    implementation's own rule, not the reference's, and differed from it whenever a rank's batch averaged less than one positive per
    image.  Outside the clamp the two are the same number.)  The totals are summed over ranks on device.
    hp["norm_avg"] (2 floats, optional): PER-IMAGE averaged normalisers of a larger virtual batch this call is one part of (tests);
    the call then divides by norm_avg * images."""

    @staticmethod
    def forward(ctx, head, reg_targets, hm_targets, pos_inds, pos_count, hp):
        head = head.contiguous()
        sums = orehip.centernet_loss_sums(head, reg_targets, hm_targets, pos_inds, pos_count, hp["gamma"], hp["beta"],
                                          hp["sigmoid_clamp"], hp["ignore_high_fp"])
        local = torch.stack([sums[1], pos_count[0].to(torch.float32)])
        images = float(hp.get("images", 1))
        if hp.get("norm_avg") is not None:
            avg = hp["norm_avg"].to(local.device, torch.float32)
        else:
            norm, world = local.clone(), 1
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                world = dist.get_world_size()
                dist.all_reduce(norm)                                 # the only other exchange of the step: 2 scalars
            avg = torch.clamp(norm / world, min=1.0) / images         # den = max(total / world, 1)
        den = avg * images
        coef = torch.stack([hp["reg_weight"] / den[0], hp["pos_weight"] * hp["alpha"] / den[1],
                            hp["neg_weight"] * (1.0 - hp["alpha"]) / den[1]])
        ctx.save_for_backward(head, reg_targets, hm_targets, pos_inds, pos_count, coef)
        ctx.hp = hp
        out = torch.cat([torch.stack([coef[0] * sums[0], -coef[1] * sums[2], -coef[2] * sums[3]]), local])
        return out

    @staticmethod
    def backward(ctx, g):
        head, reg_targets, hm_targets, pos_inds, pos_count, coef = ctx.saved_tensors
        hp = ctx.hp
        d = orehip.centernet_loss_grad(head, reg_targets, hm_targets, pos_inds, pos_count, (coef * g[:3]).contiguous(), hp["gamma"],
                                       hp["beta"], hp["sigmoid_clamp"], hp["ignore_high_fp"])
        return d, None, None, None, None, None


CN_HP = dict(gamma=2.0, beta=4.0, sigmoid_clamp=1e-4, ignore_high_fp=0.85, alpha=0.25, pos_weight=0.5, neg_weight=0.5, reg_weight=1.0)


def centernet_losses(head, reg_targets, hm_targets, pos_inds, pos_count, hp=None, with_counts: bool = False):
    out = CenterNetLossFn.apply(head, reg_targets, hm_targets, pos_inds, pos_count, hp or CN_HP)
    return (out[:3], out[3:].detach()) if with_counts else out[:3]
