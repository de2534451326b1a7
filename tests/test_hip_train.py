"""GPU parity of the training-side kernels against the oracle and the reference-run golden vectors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_model as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oh():
    import orehip
    orehip.lib()
    return orehip


def _rand_boxes(g, n, W, H, lo=2.3, span=3.2):
    ctr = torch.rand(n, 2, generator=g) * torch.tensor([float(W), float(H)])
    wh = torch.exp(torch.rand(n, 2, generator=g) * span + lo)
    b = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).clamp(min=0)
    b[:, 2].clamp_(max=W)
    b[:, 3].clamp_(max=H)
    return b


def test_cn_targets_golden(oh, golden):
    g = golden("cn_train_targets")
    shapes = [tuple(int(v) for v in s) for s in g["shapes"]]
    o = oh.centernet_targets([torch.from_numpy(g["gt0"]), torch.from_numpy(g["gt1"])], shapes)
    n = int(o["pos_count"].item())
    assert n == len(g["pos_inds"])
    assert np.array_equal(o["pos_inds"][:n].cpu().numpy(), g["pos_inds"])                 # indices: bit-exact
    assert np.array_equal(o["reg_targets"].cpu().numpy(), g["reg_targets"])               # ltrb/stride of the owning object: exact
    hm = o["hm_targets"].cpu().numpy()
    assert np.array_equal(hm == 0, g["hms"] == 0)
    assert np.abs(hm - g["hms"]).max() <= 2e-7                                            # expf: <= 2 ulp at <= 1.0


def test_cn_losses_golden(oh, golden):
    g = golden("cn_train_targets")
    dev = "cuda"
    head = torch.zeros(len(g["hm_logit"]), 16, device=dev)
    head[:, :4] = torch.from_numpy(g["reg_pred"]).to(dev)
    head[:, 4] = torch.from_numpy(g["hm_logit"]).to(dev)
    pos = torch.from_numpy(g["pos_inds"]).to(dev)
    cnt = torch.tensor([len(pos)], dtype=torch.int32, device=dev)
    s = oh.centernet_loss_sums(head, torch.from_numpy(g["reg_targets"]).to(dev), torch.from_numpy(g["hms"]).to(dev), pos, cnt).cpu()
    npos = max(len(pos), 1)
    loc = 1.0 * float(s[0]) / max(float(s[1]), 1.0)
    lpos = 0.5 * 0.25 * (-float(s[2])) / npos
    lneg = 0.5 * 0.75 * (-float(s[3])) / npos
    for got, ref in ((loc, g["loss_loc"]), (lpos, g["loss_pos"]), (lneg, g["loss_neg"])):
        assert abs(got - float(ref)) <= 2e-6 * abs(float(ref)), (got, float(ref))         # fp32 sums in a different order


@pytest.mark.parametrize("B,n_obj,hw", [(1, 40, (640, 640)), (3, 128, (512, 704)), (2, 0, (320, 320)), (2, 1, (64, 96))])
def test_cn_targets_vs_oracle(oh, B, n_obj, hw):
    g = torch.Generator().manual_seed(100 + B + n_obj)
    H, W = hw
    shapes = [(H // s, W // s) for s in (8, 16, 32)]
    gts = [_rand_boxes(g, n_obj if i != 1 else max(n_obj // 2, 0), W, H) for i in range(B)]
    if n_obj >= 40:                     # degenerate cases: box centred exactly on a grid point / on a cell border, duplicate boxes
        gts[0][0] = torch.tensor([100.0, 100.0, 140.0, 140.0])
        gts[0][1] = torch.tensor([96.0, 96.0, 160.0, 160.0])
        gts[0][2] = gts[0][1].clone()
    pos, reg, hm = R.centernet_targets(gts, shapes)
    o = oh.centernet_targets(gts, shapes)
    n = int(o["pos_count"].item())
    assert n == len(pos) and torch.equal(o["pos_inds"][:n].cpu(), pos)
    assert torch.equal(o["reg_targets"].cpu(), reg)
    hg = o["hm_targets"].cpu()
    assert torch.equal(hg == 0, hm == 0) or (hg - hm).abs().max() < 1.1e-4              # the 1e-4 cut may flip within 1 ulp
    assert (hg - hm).abs().max() <= 1.1e-4 and ((hg - hm).abs() > 2e-7).sum() <= 2


def test_cn_losses_vs_oracle(oh):
    g = torch.Generator().manual_seed(5)
    H, W, B = 640, 640, 2
    shapes = [(H // s, W // s) for s in (8, 16, 32)]
    gts = [_rand_boxes(g, 30, W, H) for _ in range(B)]
    pos, reg, hm = R.centernet_targets(gts, shapes)
    M = reg.shape[0]
    reg_pred = torch.relu(torch.randn(M, 4, generator=g) * 2 + 3)
    logit = torch.randn(M, generator=g) * 3 - 2
    ref = R.centernet_losses(reg_pred, logit, pos, reg, hm)["sums"]
    head = torch.zeros(M, 16)
    head[:, :4] = reg_pred
    head[:, 4] = logit
    s = oh.centernet_loss_sums(head.cuda(), reg.cuda(), hm.cuda(), pos.cuda(), torch.tensor([len(pos)], dtype=torch.int32).cuda()).cpu()
    assert float(s[1]) == float(ref[1])
    for k in (0, 2, 3):
        assert abs(float(s[k]) - float(ref[k])) <= 3e-6 * abs(float(ref[k])), (k, float(s[k]), float(ref[k]))
    # repeatable bit for bit (fixed-order reduction)
    s2 = oh.centernet_loss_sums(head.cuda(), reg.cuda(), hm.cuda(), pos.cuda(), torch.tensor([len(pos)], dtype=torch.int32).cuda()).cpu()
    assert torch.equal(s, s2)


def test_sgd_step_matches_torch_optim(oh):
    """ore_sgd_step_fwd == clip_grad_value_ + torch.optim.SGD(momentum, per-group lr / weight decay) over 3 steps."""
    import torch.nn as nn
    from fewx.solver.build import FlatBucket, FlatSGD
    torch.manual_seed(3)
    shapes = [(300, 37), (300,), (5, 129), (129, 3, 3, 3), (1,)]
    cpu_p = [nn.Parameter(torch.randn(*s)) for s in shapes]
    lrs, wds = [0.01, 0.01, 0.02, 0.01, 0.02], [1e-4, 0.0, 1e-4, 1e-2, 1e-4]
    ref = torch.optim.SGD([{"params": [p], "lr": lr, "weight_decay": wd} for p, lr, wd in zip(cpu_p, lrs, wds)], 0.01, momentum=0.9)
    gpu_p = [nn.Parameter(p.detach().clone().cuda()) for p in cpu_p]
    bucket = FlatBucket([("p%d" % i, p, lr, wd) for i, (p, lr, wd) in enumerate(zip(gpu_p, lrs, wds))], n_slices=2, min_slice_bytes=1024)
    opt = FlatSGD(bucket, 0.01, 0.9, 1.0)
    by_name = dict(zip(bucket.names, bucket.tensors))
    for step in range(3):
        f = [0.5, 1.0, 0.1][step]
        opt.set_lr_factor(f)
        for g, lr in zip(ref.param_groups, lrs):
            g["lr"] = lr * f
        opt.zero_grad()
        for i, p in enumerate(cpu_p):
            gr = torch.randn(p.shape) * 3.0                      # many entries beyond the clip value
            p.grad = gr.clone()
            by_name["p%d" % i].grad.add_(2.0 * gr.cuda())        # "sum over 2 ranks", averaged by grad_scale
        bucket.grad_scale = 0.5
        torch.nn.utils.clip_grad_value_(cpu_p, 1.0)
        ref.step()
        opt.step()
        for i, p in enumerate(cpu_p):
            got = by_name["p%d" % i].detach().cpu()
            assert (got - p.detach()).abs().max() <= 2e-7 * max(1.0, float(p.detach().abs().max())), (step, i)
    assert float(bucket.params[bucket.offsets[1] + 300:bucket.offsets[1] + 512].abs().max()) == 0.0      # padding never moves


# ---------------------------------------------------------------------------------------------------------------------------
# backward kernels through the autograd bindings vs torch CPU autograd of the same op (fp32; tolerance 2e-5 of the tensor's max)
# ---------------------------------------------------------------------------------------------------------------------------
def _close(got, ref, tol=2e-5):
    got, ref = got.detach().cpu().float(), ref.detach().float()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = float((got - ref).abs().max())
    scale = max(float(ref.abs().max()), 1e-6)
    assert err <= tol * scale, (err, scale)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,H,W,OH,OW,C", [(3, 30, 30, 32, 32, 16), (2, 15, 15, 16, 16, 8), (2, 32, 32, 1, 3, 128), (2, 16, 16, 3, 1, 128),
                                           (1, 8, 8, 1, 1, 128), (2, 7, 11, 3, 5, 4), (1, 5, 5, 9, 12, 4)])
def test_adaptive_avg_pool_nhwc_and_group_mean(oh, B, H, W, OH, OW, C):
    """ore_adaptive_avgpool_nhwc_fwd / _bwd against F.adaptive_avg_pool2d (value and gradient, up- and down-sampling bins), and the
    mean over an image's shots (ore_group_mean_*)."""
    from orehip import autograd as A
    g = torch.Generator().manual_seed(H + OH + C)
    x = torch.randn(B, C, H, W, generator=g)
    up = torch.randn(B, C, OH, OW, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.adaptive_avg_pool2d(xr, (OH, OW))
    (ref * up).sum().backward()
    xg = _nhwc(x).cuda().requires_grad_(True)
    y = A.adaptive_avg_pool(xg, OH, OW)
    _close(y.permute(0, 3, 1, 2), ref)
    (y * _nhwc(up).cuda()).sum().backward()
    _close(xg.grad.permute(0, 3, 1, 2), xr.grad)
    # group mean: 3 groups of B members
    z = torch.randn(3 * B, H, W, C, generator=g)
    zr = z.clone().requires_grad_(True)
    mref = zr.reshape(3, B, H, W, C).mean(1)
    um = torch.randn(3, H, W, C, generator=g)
    (mref * um).sum().backward()
    zg = z.cuda().requires_grad_(True)
    m = A.group_mean(zg, 3)
    _close(m, mref)
    (m * um.cuda()).sum().backward()
    _close(zg.grad, zr.grad)


@pytest.mark.parametrize("B,H,W,G,S", [(3, 32, 32, 32, 4), (2, 16, 16, 16, 8), (5, 8, 8, 8, 16), (1, 6, 10, 4, 4)])
def test_sm_block_glue_kernels(oh, B, H, W, G, S):
    """The SM_Block's training-side glue as HIP kernels: the two mixing layouts and their inverses (granule transposes) are exactly the
    reference's permutes; mean-pair and the re-weighted sum match the torch expressions, values and gradients."""
    from orehip import autograd as A
    g = torch.Generator().manual_seed(B + H + G)
    C = G * S
    x = torch.randn(B, H, W, C, generator=g).cuda()
    dims = (B, H, W, G, S)
    x5 = x.reshape(B, H, W, G, S)
    ph = oh.sm_permute(x, *dims, "h", False)
    pw = oh.sm_permute(x, *dims, "w", False)
    assert torch.equal(ph, x5.permute(0, 3, 2, 1, 4).contiguous()) and torch.equal(pw, x5.permute(0, 3, 1, 2, 4).contiguous())
    assert torch.equal(oh.sm_permute(ph, *dims, "h", True), x) and torch.equal(oh.sm_permute(pw, *dims, "w", True), x)
    # autograd of the layout change
    xr = x.clone().requires_grad_(True)
    up = torch.randn(ph.shape, generator=g).cuda()
    (A.sm_permute(xr, dims, "h") * up).sum().backward()
    assert torch.equal(xr.grad, up.permute(0, 3, 2, 1, 4).reshape(B, H, W, C))
    # mean-pair + combine against torch, forward and backward
    h0, w0 = torch.randn(B, H, W, C, generator=g).cuda(), torch.randn(B, H, W, C, generator=g).cuda()
    a0_, a1_ = torch.rand(B, C, generator=g).cuda(), torch.rand(B, C, generator=g).cuda()
    uy, um = torch.randn(B, H, W, C, generator=g).cuda(), torch.randn(B, C, generator=g).cuda()
    outs = []
    for hip in (False, True):
        h, w, a0, a1 = (t.clone().requires_grad_(True) for t in (h0, w0, a0_, a1_))
        if hip:
            m, y = A.mean_pair(h, w), A.combine2(w, h, a0, a1)
        else:
            m = (h + w).permute(0, 3, 1, 2).flatten(2).mean(2)
            y = w * a0[:, None, None, :] + h * a1[:, None, None, :]
        ((y * uy).sum() + (m * um).sum()).backward()
        outs.append((m, y, h.grad, w.grad, a0.grad, a1.grad))
    for got, ref in zip(outs[1], outs[0]):
        _close(got, ref.cpu(), tol=2e-5)
    # both layouts from one node, the two gradients back into one tensor (the second transpose accumulates)
    xr2 = x.clone().requires_grad_(True)
    uh, uw = torch.randn(ph.shape, generator=g).cuda(), torch.randn(pw.shape, generator=g).cuda()
    qh, qw = A.sm_dual_permute(xr2, dims)
    assert torch.equal(qh, ph) and torch.equal(qw, pw)
    ((qh * uh).sum() + (qw * uw).sum()).backward()
    want = uh.permute(0, 3, 2, 1, 4).reshape(B, H, W, C) + uw.permute(0, 2, 3, 1, 4).reshape(B, H, W, C)
    _close(xr2.grad, want.cpu(), tol=1e-6)
    # the tail node (pooled mean -> re-weighting MLP -> softmax -> re-weighted sum) against the same expression in torch ops
    torch.manual_seed(3)
    mlp = torch.nn.Sequential(torch.nn.Linear(C, C // 2), torch.nn.GELU(), torch.nn.Linear(C // 2, 2 * C)).cuda()
    outs = []
    for hip in (False, True):
        mlp.zero_grad(set_to_none=True)
        h, w = (t.clone().requires_grad_(True) for t in (h0, w0))
        if hip:
            y = A.sm_tail(w, h, mlp)
        else:
            a = mlp((h + w).permute(0, 3, 1, 2).flatten(2).mean(2)).reshape(B, C, 2).permute(2, 0, 1).softmax(0)
            y = w * a[0][:, None, None, :] + h * a[1][:, None, None, :]
        (y * uy).sum().backward()
        outs.append([y.detach(), h.grad, w.grad] + [p.grad.clone() for p in mlp.parameters()])
    for got, ref in zip(outs[1], outs[0]):
        _close(got, ref.cpu(), tol=5e-5)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [(1, 6, 9, 32, 16, 3), (2, 40, 44, 96, 96, 3), (1, 64, 64, 128, 128, 1), (4, 30, 30, 256, 112, 1),
                                              (1, 20, 20, 128, 16, 3),
                                              # every mix of full / half / quarter / empty 32 x 32 wave parts of the 64 x 64 block tile
                                              (3, 8, 8, 112, 112, 3), (2, 8, 8, 384, 112, 3), (1, 7, 9, 20, 36, 3), (1, 7, 9, 68, 100, 1),
                                              (1, 1, 2048, 128, 4, 1), (2, 16, 16, 544, 384, 1), (1, 16, 16, 80, 48, 3), (1, 12, 12, 36, 20, 1)])
def test_wgrad_with_fused_bias_gradient(oh, B, H, W, Cin, Cout, k):
    """ore_conv2d_wgrad_bias_fwd: weight gradient and bias gradient (column sums of dZ) from ONE launch, with and without a row split
    (the first shape runs unsplit and writes both outputs directly), against torch.autograd of F.conv2d.  The widths that are not
    multiples of 64 leave 16 x 16 tiles of a block without real channels; the waves owning them skip their matrix instructions
    (wg_live in ore_bwd.hip) and the result must not notice."""
    g = torch.Generator().manual_seed(B * 7 + H + Cin + Cout + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    dz = torch.randn(B, Cout, H, W, generator=g)
    (F.conv2d(x, w, b, 1, k // 2) * dz).sum().backward()
    gw, gb = oh.conv2d_wgrad(_nhwc(x).cuda(), _nhwc(dz).cuda(), k, want_bias=True)
    _close(gw, w.grad, tol=3e-5)
    _close(gb, b.grad, tol=3e-5)
    gw2 = oh.conv2d_wgrad(_nhwc(x).cuda(), _nhwc(dz).cuda(), k)
    assert torch.equal(gw2, gw)                                                  # the bias side does not disturb the weight gradient
    _close(oh.colsum(_nhwc(dz).cuda()), b.grad, tol=3e-5)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,mode", [(2, 13, 17, 32, 48, 3, "bn_relu"), (1, 20, 24, 128, 128, 3, "bias"),
                                                    (3, 9, 7, 96, 64, 1, "bias_relu"), (1, 11, 16, 128, 5, 3, "bias"),
                                                    (1, 40, 40, 256, 96, 3, "bn_relu")])
def test_conv_fn_backward(oh, B, H, W, Cin, Cout, k, mode):
    import torch.nn.functional as F
    from orehip import autograd as A
    g = torch.Generator().manual_seed(B * 100 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    up = torch.randn(B, Cout, H, W, generator=g)
    if mode == "bn_relu":
        ref = F.relu(F.conv2d(x, w, None, padding=k // 2) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    else:
        ref = F.conv2d(x, w, b, padding=k // 2)
        if mode == "bias_relu":
            ref = F.relu(ref)
    (ref * up).sum().backward()
    xg = _nhwc(x.detach()).cuda().requires_grad_(True)
    wg, bg = w.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    if mode == "bn_relu":
        y = A.conv(xg, wg, None, sc.cuda(), sh.cuda(), True)
    else:
        y = A.conv(xg, wg, bg, None, None, mode == "bias_relu")
    _close(y.permute(0, 3, 1, 2), ref)
    (y * _nhwc(up).cuda()).sum().backward()
    _close(xg.grad.permute(0, 3, 1, 2), x.grad)
    _close(wg.grad, w.grad)
    if mode != "bn_relu":
        _close(bg.grad, b.grad)


def test_linear_fn_backward(oh):
    import torch.nn.functional as F
    from orehip import autograd as A
    g = torch.Generator().manual_seed(9)
    for N, cin, cout, relu in ((128, 8192, 128, True), (128, 128, 2, False), (77, 64, 256, False)):
        x = torch.randn(N, cin, generator=g, requires_grad=True)
        w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).requires_grad_(True)
        b = torch.randn(cout, generator=g).requires_grad_(True)
        up = torch.randn(N, cout, generator=g)
        ref = F.linear(x, w, b)
        ref = F.relu(ref) if relu else ref
        (ref * up).sum().backward()
        xg, wg, bg = (t.detach().cuda().requires_grad_(True) for t in (x, w, b))
        y = A.linear(xg, wg, bg, relu)
        _close(y, ref)
        (y * up.cuda()).sum().backward()
        _close(xg.grad, x.grad); _close(wg.grad, w.grad); _close(bg.grad, b.grad)


def test_osa_block_backward(oh):
    """OSA stage 5 shapes on a small map: forward + all weight gradients + input gradient vs the oracle block."""
    from orehip import autograd as A
    sd = R.synth_state_dict(0)
    pre = "backbone.bottom_up.stage5.OSA5_1."
    g = torch.Generator().manual_seed(21)
    x = (torch.randn(2, 384, 10, 12, generator=g) * 0.5).requires_grad_(True)
    names = [f"{pre}layers.{i}.OSA5_1_{i}" for i in range(3)] + [f"{pre}concat.OSA5_1_concat"]
    leaf = {n + "/conv.weight": sd[n + "/conv.weight"].clone().requires_grad_(True) for n in names}
    sd2 = dict(sd); sd2.update(leaf)
    # oracle block without eSE: replicate osa_module up to the concat conv
    outs, h = [x], x
    for i in range(3):
        h = R.conv_bn_relu(h, sd2, names[i], 1, 1)
        outs.append(h)
    ref = R.conv_bn_relu(torch.cat(outs, 1), sd2, names[3], 1, 0)
    up = torch.randn(ref.shape, generator=g)
    (ref * up).sum().backward()

    def bn(n):
        s = sd[n + "/norm.weight"] * torch.rsqrt(sd[n + "/norm.running_var"] + 1e-5)
        return s.cuda(), (sd[n + "/norm.bias"] - sd[n + "/norm.running_mean"] * s).cuda()
    wg = [leaf[n + "/conv.weight"].detach().cuda().requires_grad_(True) for n in names]
    xg = _nhwc(x.detach()).cuda().requires_grad_(True)
    y = A.osa_block(xg, [(wg[i], *bn(names[i])) for i in range(4)])
    _close(y.permute(0, 3, 1, 2), ref)
    (y * _nhwc(up).cuda()).sum().backward()
    _close(xg.grad.permute(0, 3, 1, 2), x.grad)
    for i, n in enumerate(names):
        _close(wg[i].grad, leaf[n + "/conv.weight"].grad)


def test_roi_align_backward(oh):
    from orehip import autograd as A
    g = torch.Generator().manual_seed(4)
    feats = [(torch.randn(1, 16, 40 >> l, 48 >> l, generator=g)).requires_grad_(True) for l in range(3)]
    boxes = torch.tensor([[10.0, 12.0, 90.0, 70.0], [0.0, 0.0, 300.0, 280.0], [100.0, 50.0, 380.0, 318.0], [200.0, 100.0, 230.0, 140.0],
                          [5.0, 5.0, 500.0, 400.0], [-20.0, -10.0, 60.0, 50.0], [-150.0, -120.0, 520.0, 470.0]])
    ref = R.roi_pool_levels(feats, boxes, 8)                                     # [R,C,8,8]
    up = torch.randn(ref.shape, generator=g)
    (ref * up).sum().backward()
    fg = [f.detach()[0].permute(1, 2, 0).contiguous().cuda().requires_grad_(True) for f in feats]
    out = A.roi_align(fg, boxes.cuda())                                          # [R, 64*C] ordered [pos][c]
    out4 = out.reshape(len(boxes), 8, 8, 16).permute(0, 3, 1, 2)
    _close(out4, ref)
    (out4 * up.cuda()).sum().backward()
    for a, b in zip(fg, feats):
        assert b.grad is not None and float(b.grad.abs().max()) > 0
        _close(a.grad.permute(2, 0, 1)[None], b.grad, tol=1e-5)


@pytest.mark.parametrize("Cc,B,hw", [(128, 3, (80, 80)), (48, 2, (50, 38)), (16, 1, (23, 17))])
def test_roi_align_backward_tiled_vs_scatter_forms(oh, Cc, B, hw):
    """ore_roi_align_bwd_tiled (gather: one block per 16 x 16-cell tile, ROIs added in index order, no atomics) against the two scatter
    forms it replaces in the training step -- fixed-point integer atomics (ore_roi_align_bwd_det) and fp32 atomics (ore_roi_align_bwd):
    clustered ROIs (many per cell), ROIs hanging over every border, degenerate and inverted boxes, a ROI whose sampling grid exceeds the
    separable path (bins wider than 7 cells), maps whose sides are not multiples of the tile, channel counts that are not multiples of 32."""
    g = torch.Generator().manual_seed(Cc + B)
    H0, W0 = hw
    feats = [torch.empty(B, -(-H0 // (1 << l)), -(-W0 // (1 << l)), Cc, device="cuda") for l in range(3)]
    n_per = 150
    ctr = torch.rand(B * n_per, 2, generator=g) * torch.tensor([W0 * 8.0, H0 * 8.0])
    ctr[: B * n_per // 2] = ctr[:4].repeat(B * n_per // 8 + 1, 1)[: B * n_per // 2] + torch.randn(B * n_per // 2, 2, generator=g) * 6     # clusters
    wh = torch.exp(torch.rand(B * n_per, 2, generator=g) * 5.3 + 1.5)           # 4.5 .. 900 px: all three levels (224 px <-> level 4)
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    boxes[0] = torch.tensor([-300.0, -200.0, 2600.0, 2300.0])                   # bins wider than 7 cells on the coarsest level
    boxes[1] = torch.tensor([50.0, 60.0, 50.0, 60.0])                           # empty
    boxes[2] = torch.tensor([120.0, 90.0, 80.0, 40.0])                          # inverted
    boxes[3] = torch.tensor([-500.0, -500.0, -400.0, -420.0])                   # entirely outside
    boxes[4] = torch.tensor([W0 * 8.0 - 3, H0 * 8.0 - 3, W0 * 8.0 + 90, H0 * 8.0 + 70])
    img = torch.arange(B, dtype=torch.int32).repeat_interleave(n_per)
    perm = torch.randperm(B * n_per, generator=g)                               # images interleaved in the list
    boxes, img = boxes[perm].contiguous().cuda(), img[perm].contiguous().cuda()
    dout = torch.randn(B * n_per, 64 * Cc, generator=g).cuda()
    outs = {}
    saved = (oh.ROI_BWD_DETERMINISTIC, oh.ROI_BWD_MODE)
    try:
        for mode in ("tiled", "tiled2", "fixed", "atomic"):
            oh.ROI_BWD_DETERMINISTIC, oh.ROI_BWD_MODE = mode != "atomic", "tiled" if mode.startswith("tiled") else "fixed"
            outs[mode] = [d.clone() for d in oh.roi_align_bwd(dout, feats, boxes, box_image=img)]
        oh.ROI_BWD_DETERMINISTIC, oh.ROI_BWD_MODE = True, "tiled"
        base = [torch.randn(f.shape, generator=g).cuda() for f in feats]
        acc = oh.roi_align_bwd(dout, feats, boxes, dfeats=[b.clone() for b in base], box_image=img)
    finally:
        oh.ROI_BWD_DETERMINISTIC, oh.ROI_BWD_MODE = saved
    for l in range(3):
        assert torch.equal(outs["tiled"][l], outs["tiled2"][l])                 # bit-reproducible
        scale = float(outs["fixed"][l].abs().max())
        assert scale > 0
        assert float((outs["tiled"][l] - outs["fixed"][l]).abs().max()) <= 2e-6 * scale, l
        assert float((outs["tiled"][l] - outs["atomic"][l]).abs().max()) <= 4e-6 * scale, l
        assert float((acc[l] - (base[l] + outs["tiled"][l])).abs().max()) <= 1e-6 * (scale + float(base[l].abs().max()))    # given maps: added to
    # a single image without an image list (the per-image autograd node)
    one = [f[0] for f in feats]
    sel = img == 0
    a = oh.roi_align_bwd(dout[sel].contiguous(), one, boxes[sel].contiguous())
    for l in range(3):
        assert torch.equal(a[l], outs["tiled"][l][0])


@pytest.mark.parametrize("pooled", [4, 7, 14])
def test_roi_align_backward_tiled_other_pool_sizes(oh, pooled):
    """The tiled gather at pooled sizes other than the detector's 8: fewer bins than table columns (4, 7: the reference's dead 4 x 4
    branch has this shape) and the 16-bin build without the LDS stage (14) -- against the fixed-point scatter form."""
    g = torch.Generator().manual_seed(pooled)
    B, Cc, n_per = 2, 32, 60
    feats = [torch.empty(B, 44 >> l, 36 >> l, Cc, device="cuda") for l in range(3)]
    ctr = torch.rand(B * n_per, 2, generator=g) * torch.tensor([288.0, 352.0])
    wh = torch.exp(torch.rand(B * n_per, 2, generator=g) * 4.5 + 2.0)
    wh[::9] = torch.rand(len(wh[::9]), 2, generator=g) * 300 + 460       # some on the coarsest level (sqrt(area) >= 448)
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).cuda().contiguous()
    img = torch.arange(B, dtype=torch.int32).repeat_interleave(n_per).cuda()
    dout = torch.randn(B * n_per, pooled * pooled * Cc, generator=g).cuda()
    saved = (oh.ROI_BWD_DETERMINISTIC, oh.ROI_BWD_MODE)
    try:
        oh.ROI_BWD_DETERMINISTIC, oh.ROI_BWD_MODE = True, "tiled"
        a = [d.clone() for d in oh.roi_align_bwd(dout, feats, boxes, pooled=pooled, box_image=img)]
        oh.ROI_BWD_MODE = "fixed"
        b = oh.roi_align_bwd(dout, feats, boxes, pooled=pooled, box_image=img)
    finally:
        oh.ROI_BWD_DETERMINISTIC, oh.ROI_BWD_MODE = saved
    for l in range(3):
        scale = float(b[l].abs().max())
        assert scale > 0 and float((a[l] - b[l]).abs().max()) <= 2e-6 * scale, (l, float((a[l] - b[l]).abs().max()), scale)


def test_roi_losses_one_launch_vs_torch(oh):
    """ore_roi_losses_fwd (both second-stage losses and their gradients from one launch) against the element-wise torch expression it
    replaced (custom_fast_rcnn.py:52-81 with per-image 1/n_b weights): values and gradients, on a batch with an image without valid rows,
    an image without foreground and padding rows with degenerate boxes (no loss, no gradient)."""
    import torch.nn.functional as F
    from orehip import autograd as A
    from fewx.modeling.fsod.train_forward import get_deltas
    g = torch.Generator().manual_seed(9)
    B, R = 5, 96
    RT = B * R
    ctr = torch.rand(RT, 2, generator=g) * 500 + 50
    wh = torch.rand(RT, 2, generator=g) * 150 + 10
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    gt = boxes + torch.randn(RT, 4, generator=g) * 1.5         # (sides stay positive: min side 10)
    labels = (torch.rand(RT, generator=g) > 0.3).long()
    valid = torch.rand(RT, generator=g) > 0.15
    valid[2 * R:3 * R] = False                                 # an image without a single valid row
    labels[3 * R:4 * R] = 1                                    # an image without foreground
    boxes[~valid] = torch.tensor([0.0, 0.0, 8.0, 8.0])
    gt[~valid] = 0.0                                           # log(0 / 8) on rows that must not count
    weights = (10.0, 10.0, 5.0, 5.0)
    scores0 = torch.randn(RT, 2, generator=g) * 2
    deltas0 = torch.randn(RT, 4, generator=g)
    up = torch.tensor([1.7, 0.6])
    res = []
    for hip in (False, True):
        sc = scores0.clone().cuda().requires_grad_(True)
        de = deltas0.clone().cuda().requires_grad_(True)
        bx, gtf, lb, vf = boxes.cuda(), gt.cuda(), labels.cuda(), valid.cuda()
        if hip:
            lc, lbx = A.roi_losses(sc, de, bx, gtf, lb, vf, B, R, weights)
        else:
            img = torch.arange(B).repeat_interleave(R).cuda()
            n_b = vf.reshape(B, R).sum(1).clamp(min=1).float()
            w = vf.float() / (n_b * B)[img]
            lc = (F.cross_entropy(sc, lb, reduction="none") * w).sum()
            fg = (lb == 0) & vf
            zero = torch.zeros((), device="cuda")
            tgt = torch.where(fg[:, None], get_deltas(bx, gtf, weights), zero)
            lbx = (torch.where(fg[:, None], (de - tgt).abs(), zero).sum(1) * w).sum()
        (lc * up[0] + lbx * up[1]).backward()
        res.append((float(lc), float(lbx), sc.grad.clone(), de.grad.clone()))
    (c0, b0, gs0, gd0), (c1, b1, gs1, gd1) = res
    assert abs(c0 - c1) <= 2e-6 * abs(c0) and abs(b0 - b1) <= 2e-6 * abs(b0), (c0, c1, b0, b1)
    assert float((gs0 - gs1).abs().max()) <= 1e-6 * float(gs0.abs().max())
    assert torch.equal(gd0, gd1)                               # +-w or 0: no arithmetic to differ in
    assert float(gd1[~valid.cuda()].abs().max()) == 0.0 and float(gs1[~valid.cuda()].abs().max()) == 0.0


def test_centernet_loss_fn_backward(oh):
    from orehip import autograd as A
    g = torch.Generator().manual_seed(5)
    H, W, B = 320, 384, 1
    shapes = [(H // s, W // s) for s in (8, 16, 32)]
    gts = [_rand_boxes(g, 12, W, H)]
    pos, reg, hm = R.centernet_targets(gts, shapes)
    M = reg.shape[0]
    reg_pred = torch.relu(torch.randn(M, 4, generator=g) * 2 + 3).requires_grad_(True)
    logit = (torch.randn(M, generator=g) * 3 - 2).requires_grad_(True)
    ref = R.centernet_losses(reg_pred, logit, pos, reg, hm)
    w = torch.tensor([0.7, 1.3, 2.1])
    (w[0] * ref["loss_centernet_loc"] + w[1] * ref["loss_centernet_agn_pos"] + w[2] * ref["loss_centernet_agn_neg"]).backward()
    head = torch.zeros(M, 16)
    head[:, :4], head[:, 4] = reg_pred.detach(), logit.detach()
    head = head.cuda().requires_grad_(True)
    out = A.centernet_losses(head, reg.cuda(), hm.cuda(), pos.cuda(), torch.tensor([len(pos)], dtype=torch.int32).cuda())
    for i, k in enumerate(("loss_centernet_loc", "loss_centernet_agn_pos", "loss_centernet_agn_neg")):
        assert abs(float(out[i]) - float(ref[k])) <= 3e-6 * abs(float(ref[k])) + 1e-9, (k, float(out[i]), float(ref[k]))
    (out * w.cuda()).sum().backward()
    _close(head.grad[:, :4], reg_pred.grad, tol=1e-5)
    _close(head.grad[:, 4], logit.grad, tol=1e-5)
    assert float(head.grad[:, 5:].abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------------------
# the whole training iteration: CenterNet2Detector.forward (training) on the HIP kernels vs oracle/ref_train.py (torch CPU autograd)
# ---------------------------------------------------------------------------------------------------------------------------
def _train_model(shots):
    import os
    from conftest import PKG
    from fewx.config import get_cfg
    from detectron2.modeling import build_model
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(PKG, "configs", "fsod", "finetune_vovnet.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "INPUT.FS.SUPPORT_SHOT", shots])
    cfg.freeze()
    torch.manual_seed(0)
    m = build_model(cfg)
    sd = R.synth_roi_state(R.synth_state_dict(0), 0)
    sd["roi_heads.box_head.0.fc1.weight"] = sd["roi_heads.box_head.0.fc1.weight"] * 0.02
    sd["proposal_generator.centernet_head.agn_hm.bias"] = torch.full((1,), -2.0)       # more candidates than the -4.6 init
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    m.train()
    for lvl in (3, 4, 5):
        getattr(m, f"vip_p{lvl}").reweighting.drop.p = 0.0                             # the oracle has no dropout
    return m, sd, cfg


def test_train_iteration_losses_and_gradients_vs_oracle(oh):
    from oracle import ref_train as T
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod.train_forward import train_forward
    shots = 4
    m, sd, cfg = _train_model(shots)
    img, gt, sup, sbox = T.synth_train_inputs(5, (320, 384), n_gt=9, shots=shots, support_hw=112)   # seed: see the comment at the bounds
    # ---- oracle
    leaf = T.leaf_state(sd)
    g = torch.Generator().manual_seed(11)
    ref = T.train_iteration(leaf, img, gt, sup, sbox, lambda n: torch.randperm(n, generator=g))
    sum(ref["losses"].values()).backward()
    # ---- product
    inst = Instances((320, 384))
    inst.gt_boxes = Boxes(gt)
    inst.gt_classes = torch.zeros(len(gt), dtype=torch.int64)
    item = {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}
    over = {"boxes": ref["roi_boxes"], "labels": ref["roi_labels"], "gt": ref["roi_gt"]}
    losses, aux = train_forward(m, [item], return_aux=True, roi_override=over)
    # proposals: same top-k / NMS kernels as the eval path, here with the *_TRAIN thresholds on the training head
    pb, rb = aux["proposals"].cpu(), ref["proposals"]
    assert abs(len(pb) - len(rb)) <= 2, (len(pb), len(rb))
    d = (pb[:, None, :] - rb[None, :, :]).abs().amax(2).min(1)[0]       # order may swap on 1-ulp score ties: compare as sets
    assert float((d < 1e-2).float().mean()) >= 0.99, float((d < 1e-2).float().mean())
    assert int(aux["pos_count"].item()) == len(ref["pos_inds"])
    assert torch.equal(aux["pos_inds"][: len(ref["pos_inds"])].cpu(), ref["pos_inds"])
    for k, v in ref["losses"].items():
        assert abs(float(losses[k].detach()) - float(v.detach())) <= 2e-4 * max(abs(float(v.detach())), 1e-3), (k, float(losses[k]), float(v))
    sum(losses.values()).backward()
    named = dict(m.named_parameters())
    dead = set()
    worst = []
    for k, t in leaf.items():
        if not t.requires_grad:
            continue
        p = named[k]
        if t.grad is None:
            dead.add(k)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        err = float((p.grad.cpu() - t.grad).abs().max())
        scale = max(float(t.grad.abs().max()), 1e-8)
        worst.append((err / scale, k))
    # Conditioning: on this sample the oracle's own fp32 gradients differ from an fp64 run of the same code by up to 4e-2 of the
    # tensor's max for conv3 / vip_p3 / fpn_output3 (hard decisions -- ReLU masks, the 1e-4 heatmap cut, min/max picks -- flip
    # between precisions), and by ~1e-5 for the rest.  Two fp32 implementations therefore agree to ~1e-5 on most parameters and to
    # a few 1e-3 on those; the bound below is set from that, not from the kernels (each backward kernel is tested to 2e-5 above).
    errs = sorted(w[0] for w in worst)
    print('gradient errors (rel):', [(round(a, 7), b) for a, b in sorted(worst)[-10:]])
    # Which sample: tests/sensitivity_train_sample.py (profiles/r04_train_sample_sensitivity.txt) measures how far these gradients
    # move when the frozen weights are multiplied by (1 + 3e-7 n) or the image by (1 + 1e-6 n) on unchanged kernels.  On input seed 0,
    # used until round 4, up to 37 of 72 parameters move by more than 1e-3 and the worst by 4.2e-2 -- the bounds below held on it by
    # luck and stopped holding when k_conv_kd + k_conv_gd changed a summation order (either alone passes: r04_train_test_bisect.txt).
    # On input seed 5 at most 2 parameters move beyond 1e-3 and the worst by 2.6e-3, so the bounds sit above the sample's own
    # sensitivity.  Every candidate sample flips somewhere at the 1e-3 level; none is free of it.
    assert errs[len(errs) // 2] <= 1e-4, errs[len(errs) // 2]
    assert sum(1 for v in errs if v <= 1e-3) >= 0.85 * len(errs)
    assert errs[-1] <= 2e-2, sorted(worst)[-3:]
    assert dead == {"conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "roi_heads.fc2.weight", "roi_heads.fc2.bias",
                    "roi_heads.fc3.weight", "roi_heads.fc3.bias"}
    assert all(any(k.startswith(pre) for pre in m.gradless_parameter_prefixes()) for k in dead)
    print("worst relative gradient errors:", sorted(worst)[-3:])


def test_train_step_updates_match_oracle_sgd(oh):
    """fwd + bwd + flat-bucket clip/SGD (one process): parameter deltas after one step equal clip_grad_value_ + torch.optim.SGD on
    the oracle's gradients with the reference's parameter groups; dead-branch parameters do not move."""
    from oracle import ref_train as T
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod.train_forward import train_forward
    from fewx.solver import build_optimizer, param_groups_like_reference
    shots = 4
    m, sd, cfg = _train_model(shots)
    img, gt, sup, sbox = T.synth_train_inputs(0, (256, 320), n_gt=7, shots=shots, support_hw=96)      # seed: see bound (2)
    leaf = T.leaf_state(sd)
    g = torch.Generator().manual_seed(3)
    ref = T.train_iteration(leaf, img, gt, sup, sbox, lambda n: torch.randperm(n, generator=g))
    sum(ref["losses"].values()).backward()
    groups = {n: (lr, wd) for n, _, lr, wd in param_groups_like_reference(cfg, m)}
    live = [k for k, t in leaf.items() if t.requires_grad and t.grad is not None]
    ropt = torch.optim.SGD([{"params": [leaf[k]], "lr": groups[k][0], "weight_decay": groups[k][1]} for k in live], cfg.SOLVER.BASE_LR,
                           momentum=cfg.SOLVER.MOMENTUM)
    before = {k: leaf[k].detach().clone() for k in leaf}
    gmax = {k: float(leaf[k].grad.abs().max()) for k in live}           # before clipping
    torch.nn.utils.clip_grad_value_([leaf[k] for k in live], cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE)
    ropt.step()
    opt = build_optimizer(cfg, m)
    assert sorted(opt.bucket.names) == sorted(live)                      # the bucket holds exactly the parameters that get gradients
    assert 4 * sum(opt.bucket.numels) == 4 * 4086478                      # 16.3 MB exchanged per step
    inst = Instances((256, 320))
    inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
    item = {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}
    over = {"boxes": ref["roi_boxes"], "labels": ref["roi_labels"], "gt": ref["roi_gt"]}
    losses = train_forward(m, [item], roi_override=over)
    opt.zero_grad()
    sum(losses.values()).backward()
    named = dict(m.named_parameters())
    own = {k: named[k].grad.detach().cpu().clone() for k in live}        # this path's gradients, before the step clips them in place
    opt.step()
    clip = cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE
    for k in leaf:
        if k not in named:
            continue
        d_ref = leaf[k].detach() - before[k]
        d_got = named[k].detach().cpu() - before[k]
        if k in live:
            assert float(d_ref.abs().max()) > 0
            ulp = 1.2e-7 * float(before[k].abs().max())                 # the update is rounded into the fp32 parameter
            # (1) the step's arithmetic, on this path's own gradients: clip_grad_value_ + first SGD step with the reference's groups
            lr, wd = groups[k]
            d_own = -lr * (own[k].clamp(-clip, clip) + wd * before[k])
            assert float((d_got - d_own).abs().max()) <= 1e-6 * float(d_own.abs().max()) + 2 * ulp + 1e-12, k
            # (2) against the oracle's step, in gradient terms (a clipped update hides the gradient's scale): on this sample (input
            # seed 0) perturbations at rounding level move a gradient by up to 4.0e-3 of its maximum (2 parameters; on seed 1, used
            # until round 4, 6.9e-3 on up to 9 -- profiles/r04_train_sample_sensitivity.txt); the bound is three times that.
            assert float((d_got - d_ref).abs().max()) <= lr * min(clip, 1.2e-2 * gmax[k]) + 2 * ulp + 1e-9, k
        else:
            assert float(d_got.abs().max()) == 0.0, k
    # a second forward sees the updated weights (packed layouts are rebuilt after the step)
    l2 = train_forward(m, [item], roi_override=over)
    assert abs(float(sum(l2.values()).detach()) - float(sum(losses.values()).detach())) > 0


def test_eval_after_optimizer_step_uses_the_new_weights(oh):
    """eval -> one FlatSGD step (raw-pointer HIP kernel) -> eval in the same process: the cached engine / hipGraph / packed and composed
    weights are rebuilt (FlatSGD.step bumps the parameters' version counters), the outputs change, and they equal those of a model
    freshly built from the updated state_dict.  Also: packed() never caches a temporary (the concatenated head weight)."""
    from oracle import ref_train as T
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod.train_forward import train_forward
    from fewx.solver import build_optimizer
    from orehip import autograd as A
    shots = 4
    m, sd, cfg = _train_model(shots)
    g = torch.Generator().manual_seed(9)
    sup = R.synth_support(0)
    support = {**{k: {0: v} for k, v in sup.items()}, "rcnn_8": {0: torch.randn(shots, 128, 8, 8, generator=g) * 0.1},
               "rcnn_4": {0: torch.zeros(shots, 128, 4, 4)}}
    m.set_support_dict(support)
    q = R.synth_image(5, 256, 320)

    def run(model):
        model.eval()
        with torch.no_grad():
            out = model([{"image": q, "height": 256, "width": 320}])[0]["instances"]
        e = model.engine()
        return out.scores.clone().cpu(), out.pred_boxes.tensor.clone().cpu(), e.buffer("p3", (1, 32, 40)).clone().cpu()
    s0, b0, p0 = run(m)
    e0 = m._engine
    # one training step with a large learning rate
    m.train()
    opt = build_optimizer(cfg, m)
    opt.set_lr_factor(50.0)
    img, gt, sup_i, sbox = T.synth_train_inputs(1, (256, 320), n_gt=7, shots=shots, support_hw=96)
    inst = Instances((256, 320))
    inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
    versions = [p._version for p in opt.bucket.tensors]
    losses = train_forward(m, [{"image": img, "instances": inst, "support_images": sup_i, "support_bboxes": sbox.numpy()}])
    opt.zero_grad()
    sum(losses.values()).backward()
    opt.step()
    assert all(p._version > v for p, v in zip(opt.bucket.tensors, versions))
    assert all(len(p.__dict__.get("_ore_packed", {})) <= 4 for p in m.parameters())   # the packed copies live on the parameters (no global table)
    s1, b1, p1 = run(m)
    assert m._engine is not e0, "the engine must be rebuilt after the parameters changed"
    assert float((p1 - p0).abs().max()) > 1e-3 * float(p0.abs().max()), "FPN output did not move after a step with lr x50"
    # a model built from scratch with the updated weights gives the same answer
    from detectron2.modeling import build_model
    m2 = build_model(cfg)
    m2.load_state_dict(m.state_dict())
    m2.set_support_dict(support)
    s2, b2, p2 = run(m2)
    assert torch.equal(p1, p2) and torch.equal(s1, s2) and torch.equal(b1, b2)
    # an in-place edit in eval mode, no optimizer involved: the next call launches on the cached engine, finds it stale in the shadow
    # of that pass (ore_engine_detect_begin / _end), drops its result and repeats the pass on a rebuilt engine
    e1 = m._engine
    with torch.no_grad():
        dict(m.named_parameters())["backbone.fpn_output3.weight"].mul_(1.5)
    s3, b3, p3 = run(m)
    assert m._engine is not e1 and not torch.equal(p3, p1)
    m3 = build_model(cfg)
    m3.load_state_dict(m.state_dict())
    m3.set_support_dict(support)
    s4, b4, p4 = run(m3)
    assert torch.equal(p3, p4) and torch.equal(s3, s4) and torch.equal(b3, b4)
    # _end with no pass pending is an argument error, not a wait
    import ctypes
    import orehip
    n = ctypes.c_int32(0)
    assert orehip.lib().ore_engine_detect_end(m._engine._h, None, ctypes.byref(n)) == -22      # ORE_EINVAL
    # ... and a second _begin while a pass is pending is refused (its record would be overwritten under the running pass)
    e = m._engine
    rec = e.detect_begin(q, 256, 320)
    rec2 = torch.empty_like(rec)
    rc = orehip.lib().ore_engine_detect_begin(e._h, ctypes.c_void_p(q.data_ptr()), int(q.dtype == torch.uint8), 256, 320, 256, 320,
                                              ctypes.c_void_p(rec2.data_ptr()), None)
    assert rc == -22
    b5, s5, c5 = e.detect_end(rec)
    assert torch.equal(s5.cpu(), s3) and torch.equal(b5.cpu(), b3)


def test_correlation_fn_backward(oh):
    """HIP depthwise correlation (forward + both backward passes) vs torch autograd of the oracle's depthwise convs."""
    import torch.nn.functional as F
    from orehip import autograd as A
    g = torch.Generator().manual_seed(8)
    for (H, W, s) in ((20, 24, 16), (7, 5, 8)):
        C = 128
        q = torch.randn(1, C, H, W, generator=g).requires_grad_(True)
        proto = (torch.randn(1, C, s, s, generator=g) * 0.7).requires_grad_(True)
        w3 = (torch.randn(C, 2 * C, 1, 1, generator=g) / 16).requires_grad_(True)
        b3 = torch.randn(C, generator=g).requires_grad_(True)
        ref = R.correlation(q, proto, w3, b3)
        up = torch.randn(ref.shape, generator=g)
        (ref * up).sum().backward()
        qg = _nhwc(q.detach()).cuda().requires_grad_(True)
        pg = proto.detach().cuda().requires_grad_(True)
        wg, bg = w3.detach().cuda().requires_grad_(True), b3.detach().cuda().requires_grad_(True)
        k11 = F.adaptive_avg_pool2d(pg, (1, 1))[0, :, 0, 0]
        k13 = F.adaptive_avg_pool2d(pg, (1, 3))[0, :, 0, :]
        k31 = F.adaptive_avg_pool2d(pg, (3, 1))[0, :, :, 0]
        y = A.conv(A.correlation_cat(qg, k11, k13, k31), wg, bg, None, None, True)
        _close(y.permute(0, 3, 1, 2), ref)
        (y * _nhwc(up).cuda()).sum().backward()
        _close(qg.grad.permute(0, 3, 1, 2), q.grad)
        _close(pg.grad, proto.grad)
        _close(wg.grad, w3.grad)
    # a training batch: every image with its OWN support kernels, one launch per kernel; each image bitwise the single-image call
    B, C, H, W = 3, 128, 20, 24
    q = torch.randn(B, H, W, C, generator=g).cuda()
    k11, k13, k31 = (torch.randn(B, C, generator=g) * 0.7).cuda(), (torch.randn(B, C, 3, generator=g) * 0.5).cuda(), (torch.randn(B, C, 3, generator=g) * 0.5).cuda()
    up = torch.randn(B, H, W, 2 * C, generator=g).cuda()
    leaves = [t.clone().requires_grad_(True) for t in (q, k11, k13, k31)]
    y = A.correlation_cat(*leaves)
    (y * up).sum().backward()
    for b in range(B):
        one = [t[b:b + 1].clone().requires_grad_(True) if i == 0 else t[b].clone().requires_grad_(True) for i, t in enumerate((q, k11, k13, k31))]
        yb = A.correlation_cat(*one)
        (yb * up[b:b + 1]).sum().backward()
        assert torch.equal(yb[0], y.detach()[b])
        assert torch.equal(one[0].grad[0], leaves[0].grad[b])
        for i in (1, 2, 3):
            assert torch.equal(one[i].grad, leaves[i].grad[b]), (b, i)


# ---------------------------------------------------------------------------------------------------------------------------
# two ranks, the real detector: flat-bucket exchange (hooks fired by the HIP backward) + fused SGD.  Both ranks share cuda:0 and
# talk over gloo (RCCL refuses two ranks on one device); the exchange logic is backend-agnostic torch.distributed.
# ---------------------------------------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "faster-orefsdet_amd"))
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist
    from oracle import ref_model as R2
    from oracle import ref_train as T
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod.train_forward import train_forward
    from fewx.solver import FlatDataParallel, build_optimizer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        shots = 4
        m, sd, cfg = _train_model(shots)
        img, gt, sup, sbox = T.synth_train_inputs(10 + rank, (256, 320), n_gt=6, shots=shots, support_hw=96)
        inst = Instances((256, 320))
        inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
        item = {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}
        opt = build_optimizer(cfg, m)
        b = opt.bucket

        def run(model_for_hooks=None):
            g = torch.Generator().manual_seed(77)
            losses = train_forward(m, [item], perm=lambda n: torch.randperm(n, generator=g))
            opt.zero_grad()
            sum(losses.values()).backward()
        run()                                                       # local gradients, no exchange
        g_local, p0 = b.grads.clone(), b.params.clone()
        dp = FlatDataParallel(m, cfg)                               # same bucket; registers the post-accumulate hooks
        assert dp.bucket is b and len(b.slices) >= 2
        run()                                                       # hooks issue the slice all-reduces during backward
        opt.step()
        gl = [torch.zeros_like(g_local) for _ in range(world)]
        dist.all_gather(gl, g_local)
        exp_p, exp_m = p0.cpu().clone(), torch.zeros_like(p0).cpu()
        R2.sgd_step_flat(exp_p, sum(t.cpu() for t in gl), exp_m, b.chunk_lr.cpu(), b.chunk_wd.cpu(), 1.0, cfg.SOLVER.MOMENTUM,
                         cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE, 1.0 / world)
        got = b.params.cpu()
        d_ref = (exp_p - p0.cpu())
        err = float((got - exp_p).abs().max())
        q.put((rank, err, float(d_ref.abs().max()), got.numpy().tobytes()[:1 << 20], float((g_local - gl[1 - rank]).abs().max())))
    finally:
        dist.destroy_process_group()


def test_two_rank_detector_train_step_flat_bucket(oh):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=400) for _ in range(2)), key=lambda t: t[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    (_, e0, d0, b0, diff0), (_, e1, d1, b1, _) = res
    assert diff0 > 0                                   # the ranks really had different gradients
    assert b0 == b1                                    # and end the step bit-identical
    assert d0 > 0 and e0 <= 2e-3 * d0 + 1e-7 and e1 <= 2e-3 * d1 + 1e-7, (e0, d0, e1, d1)


# ---------------------------------------------------------------------------------------------------------------------------
# RCCL readiness on the ONE GPU a round has: backend "nccl" (= RCCL) with world_size 1, the exchange machinery
# forced on.  A one-rank SUM is the identity, so a wrapped step must equal the unwrapped step bit for bit; what the test adds is
# that the RCCL code path RUNS: init_process_group("nccl"), the slice all-reduces issued from the backward hooks on RCCL's stream
# (transport "torch") or through the C-ABI ore_allreduce_grads on the exchange stream (transport "rccl"), the waits in front of
# the raw-pointer SGD kernel, and no AccumulateGrad stream-mismatch warning with the dense part replayed as hipGraphs.
# ---------------------------------------------------------------------------------------------------------------------------
def _rccl_one_rank_worker(port, q, transport, graph):
    import os
    import sys
    import warnings
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p_ in (root, os.path.join(root, "faster-orefsdet_amd"), os.path.join(root, "tests")):
        sys.path.insert(0, p_)
    import torch.distributed as dist
    from oracle import ref_train as T
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod.train_forward import train_forward
    from fewx.solver import FlatDataParallel, build_optimizer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        shots = 4
        img, gt, sup, sbox = T.synth_train_inputs(10, (256, 320), n_gt=6, shots=shots, support_hw=96)
        inst = Instances((256, 320))
        inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
        item = {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}

        def steps(wrap):
            """Three optimizer steps; returns the bucket after step 1 (parameters + the gradients the optimizer consumed) and at the end."""
            torch.manual_seed(0)
            m, sd, cfg = _train_model(shots)
            m.train_graph = graph
            opt = build_optimizer(cfg, m)
            p0 = opt.bucket.params.detach().cpu().clone()
            dp = FlatDataParallel(m, cfg, transport=transport, force_exchange=True) if wrap else None
            logs, g1, p1 = [], None, None
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                for it in range(3):
                    g = torch.Generator().manual_seed(77 + it)
                    losses = train_forward(m, [item], perm=lambda n: torch.randperm(n, generator=g))
                    opt.zero_grad()
                    sum(losses.values()).backward()
                    opt.step()                                     # (the fused kernel reads the gradients, it does not rewrite them)
                    if it == 0:
                        torch.cuda.synchronize()
                        g1, p1 = opt.bucket.grads.detach().cpu().clone(), opt.bucket.params.detach().cpu().clone()
                    if dp is not None:
                        logs.append(list(dp.last_issue_log))
                torch.cuda.synchronize()
            msgs = [str(x.message) for x in w if "AccumulateGrad" in str(x.message)]
            out = opt.bucket.params.detach().cpu().clone()
            n_slices = len(opt.bucket.slices)
            if dp is not None:
                dp.close()
            return {"p0": p0, "g1": g1, "p1": p1, "p3": out, "logs": logs, "msgs": msgs, "n_slices": n_slices,
                    "gerr": m.__dict__.get("_ore_train_graph_error")}
        a, a2, b = steps(False), steps(False), steps(True)
        ar = torch.ones(4, device="cuda")
        dist.all_reduce(ar)                                        # and a plain RCCL collective of the process group itself

        def dist_(x, y):
            return float((x - y).abs().max())
        q.put({"g_scale": float(a["g1"].abs().max()), "g_noise": dist_(a["g1"], a2["g1"]), "g_diff": dist_(a["g1"], b["g1"]),
               "moved1": dist_(a["p1"], a["p0"]), "p1_noise": dist_(a["p1"], a2["p1"]), "p1_diff": dist_(a["p1"], b["p1"]),
               "finite": bool(torch.isfinite(b["p3"]).all()), "moved3": dist_(b["p3"], b["p0"]), "logs": b["logs"],
               "warnings": a["msgs"] + b["msgs"], "n_slices": b["n_slices"], "backend": dist.get_backend(), "ar": ar.cpu().tolist(),
               "graph_error": a["gerr"] or b["gerr"]})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport,graph", [("torch", False), ("rccl", False), ("torch", True), ("rccl", True)])
def test_rccl_one_rank_rehearsal(oh, transport, graph):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(port, q, transport, graph))
    p.start()
    r = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    assert r["backend"] == "nccl" and r["ar"] == [1.0] * 4
    assert r["graph_error"] is None, r["graph_error"]
    # A one-rank SUM is the identity: after ONE step the gradients the optimizer consumed and the parameters it wrote EQUAL the plain
    # step's bit for bit -- since round 5 the step is bit-reproducible (the ROIAlign backward accumulates in 64-bit fixed point with
    # integer atomics, ore_roi_align_bwd_det; rounds 1-4 used fp32 atomics and this test had to compare against run-to-run noise):
    # two plain runs agree exactly, and so do the plain and the wrapped run.
    print("rehearsal:", {k: r[k] for k in ("g_scale", "g_noise", "g_diff", "moved1", "p1_noise", "p1_diff")})
    assert r["finite"] and r["moved1"] > 1e-5 and r["moved3"] > r["moved1"] * 0.5, r
    assert r["g_noise"] == 0.0 and r["p1_noise"] == 0.0, (r["g_noise"], r["p1_noise"])
    assert r["g_diff"] == 0.0 and r["p1_diff"] == 0.0, (r["g_diff"], r["p1_diff"])
    assert len(r["logs"]) == 3 and r["n_slices"] >= 2
    for log in r["logs"]:                                          # every slice exactly once per step ...
        assert sorted(s_ for s_, _ in log) == list(range(r["n_slices"])), log
        assert sum(1 for _, from_hook in log if from_hook) >= r["n_slices"] - 1, log     # ... and from the backward hooks (overlap)
    assert not r["warnings"], r["warnings"]                        # the AccumulateGrad stream-mismatch warning is gone


def test_small_training_ops_backward(oh):
    """GroupNorm+ReLU, eSE, ceil-mode max-pool and the FPN top-down add (fused in the lateral conv) vs torch CPU autograd."""
    import torch.nn.functional as F
    from orehip import autograd as A
    g = torch.Generator().manual_seed(31)
    # --- GroupNorm(32, 128) + ReLU on one image
    x = torch.randn(1, 128, 21, 17, generator=g).requires_grad_(True)
    gam = (torch.rand(128, generator=g) + 0.5).requires_grad_(True)
    bet = (torch.randn(128, generator=g) * 0.1).requires_grad_(True)
    ref = F.relu(F.group_norm(x, 32, gam, bet, 1e-5))
    up = torch.randn(ref.shape, generator=g)
    (ref * up).sum().backward()
    xg, gg, bg = _nhwc(x.detach()).cuda().requires_grad_(True), gam.detach().cuda().requires_grad_(True), bet.detach().cuda().requires_grad_(True)
    y = A.group_norm_relu(xg, gg, bg, 32, 1e-5, True)
    _close(y.permute(0, 3, 1, 2), ref)
    (y * _nhwc(up).cuda()).sum().backward()
    _close(xg.grad.permute(0, 3, 1, 2), x.grad); _close(gg.grad, gam.grad); _close(bg.grad, bet.grad)
    # --- the same on a batch of 3 in one launch per kernel (statistics per image; 357 rows per image: chunks restart per image)
    x = (torch.randn(3, 128, 21, 17, generator=g) * torch.tensor([1.0, 3.0, 0.2]).view(3, 1, 1, 1)).requires_grad_(True)
    gam.grad = bet.grad = None
    ref = F.relu(F.group_norm(x, 32, gam, bet, 1e-5))
    up = torch.randn(ref.shape, generator=g)
    (ref * up).sum().backward()
    xg, gg, bg = _nhwc(x.detach()).cuda().requires_grad_(True), gam.detach().cuda().requires_grad_(True), bet.detach().cuda().requires_grad_(True)
    y = A.group_norm_relu(xg, gg, bg, 32, 1e-5, True)
    _close(y.permute(0, 3, 1, 2), ref)
    (y * _nhwc(up).cuda()).sum().backward()
    _close(xg.grad.permute(0, 3, 1, 2), x.grad); _close(gg.grad, gam.grad); _close(bg.grad, bet.grad)
    y1 = A.group_norm_relu(xg.detach()[1:2].contiguous(), gg.detach(), bg.detach(), 32, 1e-5, True)
    assert torch.equal(y1[0], y.detach()[1])                                 # a batched image is bitwise the single-image call
    # --- eSE on a batch of 3
    C = 96
    x = torch.randn(3, C, 10, 12, generator=g).requires_grad_(True)
    w = (torch.randn(C, C, 1, 1, generator=g) / C ** 0.5).requires_grad_(True)
    b = (torch.randn(C, generator=g) * 2.0).requires_grad_(True)            # spread so both hsigmoid clamps are exercised
    ref = x * (F.relu6(F.conv2d(F.adaptive_avg_pool2d(x, 1), w, b) + 3.0) / 6.0)
    up = torch.randn(ref.shape, generator=g)
    (ref * up).sum().backward()
    xg, wg, bg = _nhwc(x.detach()).cuda().requires_grad_(True), w.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    y = A.ese(xg, wg, bg)
    _close(y.permute(0, 3, 1, 2), ref)
    (y * _nhwc(up).cuda()).sum().backward()
    _close(xg.grad.permute(0, 3, 1, 2), x.grad); _close(wg.grad, w.grad); _close(bg.grad, b.grad)
    # --- MaxPool2d(3, 2, ceil_mode=True), odd and even sizes, ties (quantised values)
    for (H, W) in ((40, 40), (13, 18), (7, 7)):
        x = (torch.randint(0, 4, (2, 16, H, W), generator=g).float()).requires_grad_(True)
        ref = F.max_pool2d(x, 3, 2, ceil_mode=True)
        up = torch.randn(ref.shape, generator=g)
        (ref * up).sum().backward()
        xg = _nhwc(x.detach()).cuda().requires_grad_(True)
        y = A.maxpool(xg)
        _close(y.permute(0, 3, 1, 2), ref)
        (y * _nhwc(up).cuda()).sum().backward()
        _close(xg.grad.permute(0, 3, 1, 2), x.grad)
    # --- lateral 1x1 conv + nearest-2x top-down add (odd size: the upsampled map is cropped)
    f = torch.randn(1, 64, 9, 14, generator=g).requires_grad_(True)
    top = torch.randn(1, 32, 5, 7, generator=g).requires_grad_(True)
    w = (torch.randn(32, 64, 1, 1, generator=g) / 8).requires_grad_(True)
    b = torch.randn(32, generator=g).requires_grad_(True)
    ref = F.conv2d(f, w, b) + F.interpolate(top, scale_factor=2, mode="nearest")[:, :, :9, :14]
    up = torch.randn(ref.shape, generator=g)
    (ref * up).sum().backward()
    fg, tg = _nhwc(f.detach()).cuda().requires_grad_(True), _nhwc(top.detach()).cuda().requires_grad_(True)
    wg, bg = w.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    y = A.conv(fg, wg, bg, None, None, False, tg)
    _close(y.permute(0, 3, 1, 2), ref)
    (y * _nhwc(up).cuda()).sum().backward()
    _close(fg.grad.permute(0, 3, 1, 2), f.grad); _close(tg.grad.permute(0, 3, 1, 2), top.grad); _close(wg.grad, w.grad); _close(bg.grad, b.grad)


def test_graph_captured_dense_part_matches_eager(oh):
    """model.train_graph = True replays the shape-static dense part (forward and backward) as hipGraphs: losses, gradients and
    the parameters after two optimizer steps must equal the eager path (same kernels, same order)."""
    from oracle import ref_train as T
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod.train_forward import train_forward
    from fewx.solver import build_optimizer
    shots = 4
    img, gt, sup, sbox = T.synth_train_inputs(2, (256, 320), n_gt=6, shots=shots, support_hw=96)
    res = {}
    for mode in ("eager", "graph"):
        m, sd, cfg = _train_model(shots)
        m.train_graph = mode == "graph"
        opt = build_optimizer(cfg, m)
        # the synthetic model trains chaotically at the config's rate (value-clipped SGD moves EVERY weight by +-lr per step, the second
        # stage's loss explodes by step 2 and the fp32-atomics order of the ROIAlign backward is amplified to percents -- eager vs eager
        # as much as eager vs graph): a damped rate keeps three steps comparable while every step still changes all the parameters
        opt.set_lr_factor(0.02)
        inst = Instances((256, 320))
        inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
        item = {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}
        out = []
        for step in range(3):
            g = torch.Generator().manual_seed(5 + step)
            losses = train_forward(m, [item], perm=lambda n: torch.randperm(n, generator=g))
            opt.zero_grad()
            sum(losses.values()).backward()
            out.append(({k: float(v.detach()) for k, v in losses.items()}, opt.bucket.grads.clone()))
            opt.step()
        assert m.__dict__.get("_ore_train_graph_error") is None, m.__dict__.get("_ore_train_graph_error")
        res[mode] = (out, opt.bucket.params.clone())
    for step in range(3):
        le, ge = res["eager"][0][step]
        lg, gg = res["graph"][0][step]
        # step 0 starts from identical parameters: same kernels, same order -> tight.  Later steps start from parameters that already
        # differ by the fp32-atomics order of the ROIAlign backward (value-clipped SGD on this synthetic model moves every weight by
        # +-lr whatever the gradient size, so a sign flip of a near-zero gradient is a full step): that noise grows step over step
        # in BOTH directions (eager run vs eager run as much as eager vs graph), hence the looser bound after step 0.
        ltol, gtol = (1e-5, 1e-4) if step == 0 else (2e-3, 2e-3)
        for k in le:
            assert abs(le[k] - lg[k]) <= ltol * max(abs(le[k]), 1e-3), (step, k, le[k], lg[k])
        assert float((ge - gg).abs().max()) <= gtol * float(ge.abs().max()), (step, float((ge - gg).abs().max()), float(ge.abs().max()))
    assert float((res["eager"][1] - res["graph"][1]).abs().max()) <= 1e-5        # ROIAlign backward uses fp32 atomics: order differs run to run


def test_whole_step_graph_matches_eager(oh, monkeypatch):
    """fewx.solver.GraphedTrainStep: forward + losses + backward + clip / SGD of one iteration captured once and replayed.  With the same
    inputs, the same sampling keys and the schedule factor changing every step, the parameters after three eager + three replayed
    iterations must be those of six eager iterations: the replay runs the same kernels in the same order, the ROIAlign backward adds in
    ROI order, the LR factor is read from the device -- equality to the last bit is the expectation, 1e-6 of the largest weight the bar."""
    from oracle import ref_train as T
    from detectron2.structures import Boxes, Instances
    from fewx.solver import GraphedTrainStep, build_optimizer
    shots, B = 4, 2
    items = []
    for b in range(B):
        img, gt, sup, sbox = T.synth_train_inputs(2 + b, (256, 320), n_gt=5 + b, shots=shots, support_hw=96)
        inst = Instances((256, 320))
        inst.gt_boxes, inst.gt_classes = Boxes(gt.cuda()), torch.zeros(len(gt), dtype=torch.int64).cuda()
        items.append({"image": img.cuda(), "instances": inst, "support_images": sup.cuda(), "support_bboxes": sbox.cuda()})
    keys = torch.rand(B, 16384, generator=torch.Generator().manual_seed(3)).cuda()
    real_rand = torch.rand

    def fixed_rand(*shape, **kw):                           # the fg / bg subsample draws its keys here: the same for both runs
        if len(shape) == 2 and kw.get("device") is not None and torch.device(kw["device"]).type == "cuda":
            return keys[:shape[0], :shape[1]].clone()
        return real_rand(*shape, **kw)

    monkeypatch.setattr(torch, "rand", fixed_rand)
    res = {}
    for mode in ("eager", "graph"):
        m, sd, cfg = _train_model(shots)
        opt = build_optimizer(cfg, m)
        stepper = GraphedTrainStep(m, opt, warmup=3) if mode == "graph" else None
        losses_log = []
        for it in range(6):
            opt.set_lr_factor(0.02 * (1.0 + 0.25 * it))     # a schedule that moves every iteration
            if stepper is not None:
                losses = stepper(items)
            else:
                losses = m(items)
                opt.zero_grad()
                sum(losses.values()).backward()
                opt.step()
            losses_log.append({k: float(v.detach()) for k, v in losses.items()})
        torch.cuda.synchronize()
        if stepper is not None:
            assert stepper.error is None, stepper.error
            assert stepper.eager_steps == 3 and stepper.replays == 3
        res[mode] = (losses_log, opt.bucket.params.detach().clone(), opt.bucket.momentum.detach().clone())
    for it in range(6):
        for k, v in res["eager"][0][it].items():
            assert abs(v - res["graph"][0][it][k]) <= 1e-5 * max(abs(v), 1e-3), (it, k, v, res["graph"][0][it][k])
    pe, pg = res["eager"][1], res["graph"][1]
    assert float((pe - pg).abs().max()) <= 1e-6 * float(pe.abs().max()), float((pe - pg).abs().max())
    assert float((res["eager"][2] - res["graph"][2]).abs().max()) <= 1e-5 * float(res["eager"][2].abs().max())
    # a later eval on the replay-trained model sees the new weights (the replay bumps the version counters host-side)
    assert all(p._version > 0 for p in [next(iter(m.parameters()))])


def test_centernet_normaliser_is_the_references(oh):
    """ref:fewx/modeling/fsod/fsod_rpn.py:712-716,748-751: the CenterNet sums of a rank's WHOLE batch are divided by
    max(reduce_sum(n) / num_gpus, 1).  In the clamped regime -- here two images without any ground truth, n = 0 -- the batch loss is
    therefore the SUM of the two single-image losses (each divided by 1), not their mean (what max(total / world, images), the rule of
    rounds 2-3, gave); with positives the batch loss is the positives-weighted combination and both rules coincide."""
    from oracle import ref_train as T
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod import train_forward as TF
    shots = 4
    m, sd, cfg = _train_model(shots)

    def item(seed, n_gt):
        img, gt, sup, sbox = T.synth_train_inputs(seed, (256, 320), n_gt=max(n_gt, 1), shots=shots, support_hw=96)
        gt = gt[:n_gt]
        inst = Instances((256, 320))
        inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
        return {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}
    e1, e2 = item(2, 0), item(3, 0)
    l1, l2, l12 = TF.train_forward(m, [e1]), TF.train_forward(m, [e2]), TF.train_forward(m, [e1, e2])
    k = "loss_centernet_agn_neg"
    want = float(l1[k]) + float(l2[k])
    assert float(l1[k]) > 0 and abs(float(l12[k]) - want) <= 1e-5 * want, (float(l12[k]), want)
    # with positives: sums over the batch / total positives
    a, b = item(1, 5), item(4, 3)
    la, aux_a = TF.train_forward(m, [a], return_aux=True)
    lb, aux_b = TF.train_forward(m, [b], return_aux=True)
    lab, aux = TF.train_forward(m, [a, b], return_aux=True)
    ca, cb, cab = aux_a["cn_counts"].cpu(), aux_b["cn_counts"].cpu(), aux["cn_counts"].cpu()
    assert torch.allclose(ca + cb, cab) and float(cab.min()) >= 2
    for key, idx in (("loss_centernet_loc", 0), ("loss_centernet_agn_pos", 1), ("loss_centernet_agn_neg", 1)):
        want = (float(la[key]) * float(ca[idx]) + float(lb[key]) * float(cb[idx])) / float(cab[idx])
        assert abs(float(lab[key]) - want) <= 2e-5 * max(abs(want), 1e-3), (key, float(lab[key]), want)


def test_train_forward_batch_and_empty_gt(oh):
    """A list of B images gives what B data-parallel single-image ranks of the reference give after gradient averaging (the
    reference itself returns the last image's losses for a longer list: SURVEY App. C.1); an image without ground truth trains on
    background only (finite losses, no box-regression term)."""
    from oracle import ref_train as T
    from detectron2.structures import Boxes, Instances
    shots = 4
    m, sd, cfg = _train_model(shots)

    def item(seed, n_gt):
        img, gt, sup, sbox = T.synth_train_inputs(seed, (256, 320), n_gt=max(n_gt, 1), shots=shots, support_hw=96)
        gt = gt[:n_gt]
        inst = Instances((256, 320))
        inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.ones(len(gt), dtype=torch.int64)     # forced to class 0 by the detector
        return {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}
    a, b = item(1, 5), item(2, 0)
    from fewx.modeling.fsod import train_forward as TF
    torch.manual_seed(0)
    la, lb = m([a]), m([b])
    assert all(torch.isfinite(v) for v in la.values())
    assert float(lb["loss_box_reg_stage0"]) == 0.0 and all(torch.isfinite(v) for v in lb.values())
    assert float(lb["loss_centernet_agn_pos"]) == 0.0 and float(lb["loss_centernet_loc"]) == 0.0
    assert int(a["instances"].gt_classes.sum()) == 0                                           # gt_classes forced to 0 (fsod_cen.py:158-159)
    # B images on one rank = B data-parallel ranks of the reference with one image each: the CenterNet normalisers are the all-image
    # totals / B (fsod_rpn.py:712-716: reduce_sum / num_gpus), the second-stage losses the mean over the images.
    torch.manual_seed(0)
    lab, aux = TF.train_forward(m, [a, b], return_aux=True)
    navg = (aux["cn_counts"] / 2).clamp(min=1.0)
    la1, lb1 = TF.train_forward(m, [a], cn_norm_avg=navg), TF.train_forward(m, [b], cn_norm_avg=navg)
    for k in lab:
        if k.startswith("loss_centernet"):                                                      # deterministic part (no sampling)
            want = 0.5 * (float(la1[k]) + float(lb1[k]))
            assert abs(float(lab[k]) - want) <= 1e-5 * max(abs(want), 1e-3), k
    sum(lab.values()).backward()
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in m.parameters())
    # images of different sizes in one call: ImageList.from_tensors semantics (normalise, zero-pad to the batch maximum) through the
    # generic path; one size per batch goes through the fused-preprocess stem (raw images handed to stem_1) -- same numbers
    img_d, gt_d, sup_d, sbox_d = T.synth_train_inputs(6, (224, 288), n_gt=3, shots=shots, support_hw=96)
    inst_d = Instances((224, 288))
    inst_d.gt_boxes, inst_d.gt_classes = Boxes(gt_d), torch.zeros(len(gt_d), dtype=torch.int64)
    d_item = {"image": img_d, "instances": inst_d, "support_images": sup_d, "support_bboxes": sbox_d.numpy()}
    m.zero_grad(set_to_none=True)
    lmix = TF.train_forward(m, [a, d_item])
    sum(lmix.values()).backward()
    assert all(torch.isfinite(v) for v in lmix.values()) and all(p.grad is None or torch.isfinite(p.grad).all() for p in m.parameters())
    over = {"boxes": aux["roi_boxes"][0], "labels": aux["roi_labels"][0], "gt": aux["roi_gt"][0]}
    l_raw = TF.train_forward(m, [a], roi_override=over)
    l_gen = TF.train_forward(m, [a], roi_override=over, fused_preprocess=False)
    for k in l_raw:
        assert abs(float(l_raw[k]) - float(l_gen[k])) <= 2e-5 * max(abs(float(l_gen[k])), 1e-3), k
    # the batched pass (two batched backbone passes, ONE second-stage pass over the ROIs of both images) against the two
    # single-image passes, with a deterministic fg/bg subsample so that all five losses and the gradients are comparable
    det = lambda n: torch.arange(n - 1, -1, -1)                                                 # noqa: E731
    # The two kinds of pass run different conv kernels on the frozen stages (the batch crosses the Winograd row threshold): features
    # differ by ~1e-6, harmless everywhere except AT the two kinks of the trainable eSE gates' hsigmoid, where d gate / d z jumps
    # between 0 and 1/6 (seed 3 puts a stage-5 pre-activation 2.4e-6 from z = 3, and its derivative flips between the passes).  The
    # comparison is defined where both passes sit on the same linear piece, so a candidate second image whose masks differ is skipped.
    zs = []
    real_addmm = torch.addmm

    def spy(*args, **kw):                                                                       # EseFn.forward: z = addmm(fc_b, mean, W^T)
        zs.append(real_addmm(*args, **kw).detach())
        return real_addmm(*args, **kw)

    def spied(fn):
        del zs[:]
        torch.addmm = spy
        try:
            out = fn()
        finally:
            torch.addmm = real_addmm
        return out, [((z > -3.0) & (z < 3.0)) for z in zs]
    for seed_c in (3, 4, 5, 6):
        c = item(seed_c, 3)
        m.zero_grad(set_to_none=True)
        (l2, aux), mask_b = spied(lambda: TF.train_forward(m, [a, c], perm=det, return_aux=True))
        assert len(mask_b) >= 2                                                                 # the trainable stages' gates were seen
        assert len(aux["rois_per_image"]) == 2 and int(aux["valid"].sum()) == sum(aux["rois_per_image"])
        sum(l2.values()).backward()
        batch_grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        navg = (aux["cn_counts"] / 2).clamp(min=1.0)
        singles, grads, same_piece = [], [], True
        for i, it in enumerate((a, c)):
            m.zero_grad(set_to_none=True)
            # the single pass trains on the batch's ROI sample of this image (a 1e-6 heatmap difference may swap two proposals of
            # nearly equal score)
            over_i = {"boxes": aux["roi_boxes"][i], "labels": aux["roi_labels"][i], "gt": aux["roi_gt"][i]}
            l1, mask_1 = spied(lambda: TF.train_forward(m, [it], roi_override=over_i, cn_norm_avg=navg))
            # (query pass: one row per image; support pass: `shots` rows per image)
            same_piece = same_piece and len(mask_1) == len(mask_b) and all(
                torch.equal(mb[i * m1.shape[0]:(i + 1) * m1.shape[0]], m1) for mb, m1 in zip(mask_b, mask_1))
            sum(l1.values()).backward()
            singles.append({k: float(v) for k, v in l1.items()})
            grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        if same_piece:
            break
    else:
        pytest.fail("every candidate image flips an eSE hsigmoid derivative between the batched and the single pass")
    for k in singles[0]:
        want = 0.5 * (singles[0][k] + singles[1][k])
        assert abs(float(l2[k]) - want) <= 2e-4 * max(abs(want), 1e-3), (k, float(l2[k]), want)
    errs = []
    for n, p in m.named_parameters():
        if n not in batch_grads:
            assert n not in grads[0]
            continue
        want = 0.5 * (grads[0][n] + grads[1][n])
        errs.append((float((batch_grads[n] - want).abs().max() / want.abs().max().clamp_min(1e-12)), n))
    errs.sort()
    assert errs[len(errs) // 2][0] <= 1e-4 and errs[int(len(errs) * 0.85)][0] <= 1e-3 and errs[-1][0] <= 2e-2, errs[-12:]


def test_colsum_segments_and_scaled_gate_weight(oh):
    """ore_colsum_segments_fwd: per-image column sums of a batch, each image bitwise the unsegmented call on its rows, and equal to a
    float64 sum at 1e-5.  ore_ese_gate_scaled_weight_fwd: the eSE gate of one image (vs the reference formula relu6(fc(mean) + 3) / 6,
    d2z:modeling/backbone/vovnet.py:238-260) and the gate-scaled copy of a packed 1x1 weight, exactly w * gate."""
    import orehip
    g = torch.Generator().manual_seed(23)
    S, rps, Cc = 5, 700, 224
    x = torch.randn(S * rps, Cc, generator=g).cuda()
    got = orehip.colsum_segments(x, S)
    for s_ in range(S):
        assert torch.equal(got[s_], orehip.colsum(x[s_ * rps:(s_ + 1) * rps].contiguous()))
    ref = x.double().reshape(S, rps, Cc).sum(1)
    assert float((got.double() - ref).abs().max() / ref.abs().max()) < 1e-5
    P, HW, Cc = 25, 400, 512
    part = torch.randn(P, Cc, generator=g).cuda() * 3.0
    fw, fb = (torch.randn(Cc, Cc, generator=g) / Cc ** 0.5).cuda(), (torch.randn(Cc, generator=g) * 2.0).cuda()
    wl = torch.randn(128, Cc, generator=g).cuda()
    gate, ws = orehip.ese_gate_scaled_weight(part, HW, fw, fb, wl)
    mean = part.double().sum(0) / HW
    want = torch.clamp(fw.double() @ mean + fb.double() + 3.0, 0.0, 6.0) / 6.0
    assert float((gate[0].double() - want).abs().max()) < 1e-5
    assert torch.equal(gate, orehip.ese_gate_from_colsum(part, HW, fw, fb))
    assert torch.equal(ws, wl * gate)


def test_detect_batch_equals_per_image_detect(oh):
    """ore_detect_batch_fwd (the train-mode proposals of a batch: the greedy NMS scans of all images in one launch) returns, image by
    image, exactly what ore_detect_fwd returns -- boxes, scores, keep indices and counts bit for bit (4000 / 0.9 / 2000 thresholds)."""
    import orehip
    g = torch.Generator().manual_seed(17)
    B, shapes = 5, ((40, 48), (20, 24), (10, 12))
    per_image = []
    for b in range(B):
        hs = []
        for (H, W) in shapes:
            h = torch.zeros(H, W, 16)
            h[..., :4] = torch.rand(H, W, 4, generator=g) * (3.0 + b) + 0.3
            h[..., 4] = torch.randn(H, W, generator=g) * 2.0 - (1.0 if b != 3 else 14.0)    # image 3: next to no candidates
            hs.append(h.cuda())
        per_image.append(hs)
    many = orehip.detect_batch(per_image, (8, 16, 32), 1e-5, 4000, 0.9, 2000)
    torch.cuda.synchronize()
    for b in range(B):
        one = orehip.detect(per_image[b], (8, 16, 32), 1e-5, 4000, 0.9, 2000)
        n, n_pre = int(one["counts"][1]), int(one["counts"][0])
        assert (n > 100 or b == 3) and torch.equal(one["counts"], many[b]["counts"])
        for k in ("out_boxes", "out_scores", "keep_idx"):
            assert torch.equal(one[k][:n], many[b][k][:n]), (b, k)
        for k in ("pre_boxes", "pre_scores", "pre_loc", "pre_level"):                 # (select / rank / mask of the batch share three launches)
            assert torch.equal(one[k][:n_pre], many[b][k][:n_pre]), (b, k)


def test_sample_rois_device_properties(oh):
    """The sync-free fg/bg subsample (train_forward.sample_rois_device) against label_and_sample_proposals' contract
    (d2z:modeling/roi_heads/roi_heads.py:181-295, sampling.py:10-53) on a batch with an image without ground truth and an image with
    fewer candidates than the sample size: labels follow the IoU >= 0.6 matcher, the matched gt is the arg-max, at most 64 foreground
    of 128, no candidate drawn twice, padding rows flagged; when nothing has to be dropped the sample is exactly the candidate set the
    host-shaped reference logic (label_and_sample, pinned by tests/golden/roi_train_pieces.npz) returns."""
    from fewx.modeling.fsod import train_forward as TF
    m, sd, cfg = _train_model(4)
    rh = m.roi_heads
    g = torch.Generator().manual_seed(12)
    B, cap = 3, 400
    gts = [torch.tensor([[30., 40., 130., 160.], [200., 50., 290., 140.], [100., 180., 220., 250.]]), torch.zeros(0, 4),
           torch.tensor([[50., 50., 150., 150.]])]
    prop = torch.zeros(B, cap, 4)
    n = [400, 300, 40]
    for b in range(B):
        c = torch.rand(cap, 2, generator=g) * 280 + 20
        wh = torch.rand(cap, 2, generator=g) * 100 + 20
        prop[b] = torch.cat([c - wh / 2, c + wh / 2], 1)
        for k, gb in enumerate(gts[b]):                                   # jittered copies of the gt: plenty of foreground
            prop[b, k * 30:(k + 1) * 30] = gb + torch.randn(30, 4, generator=g) * 4
    gtp, gt_n = TF._pad_stack([t.cuda() for t in gts], 4, torch.device("cuda"))
    torch.manual_seed(3)
    boxes, labels, rgt, valid = TF.sample_rois_device(rh, prop.cuda(), torch.tensor(n).cuda(), gtp, gt_n)
    assert boxes.shape == (B, 128, 4) and labels.shape == (B, 128) and valid.dtype == torch.bool
    for b in range(B):
        cand = torch.cat([prop[b, :n[b]], gts[b]], 0)
        v = valid[b].cpu()
        bx, lb, rg = boxes[b].cpu()[v], labels[b].cpu()[v], rgt[b].cpu()[v]
        nfg_all = 0
        if len(gts[b]):
            iou = TF.pairwise_iou(gts[b], cand)
            vals, midx = iou.max(0)
            nfg_all = int((vals >= 0.6).sum())
        assert int(v.sum()) == min(128, len(cand)) and int((lb == 0).sum()) == min(nfg_all, 64)
        rows = [(cand == r).all(1).nonzero()[0, 0].item() for r in bx]     # every sampled box is a candidate ...
        assert len(set(rows)) == len(rows)                                  # ... drawn once
        if len(gts[b]):
            assert torch.equal(lb, torch.where(vals[rows] >= 0.6, 0, 1)) and torch.equal(rg, gts[b][midx[rows]])
            assert torch.all(lb[: int((lb == 0).sum())] == 0)               # foreground first, like cat([pos_idx, neg_idx])
        else:
            assert torch.all(lb == 1)
        assert torch.all(labels[b].cpu()[~v] == 1)
    # the one-launch kernel against the element-wise torch form it replaced, same keys (same seed, same single draw): the same sample,
    # row for row -- boxes, labels, matched gt, padding
    torch.manual_seed(3)
    tb, tl, tg, tv = TF.sample_rois_torch(rh, prop.cuda(), torch.tensor(n).cuda(), gtp, gt_n)
    assert torch.equal(valid, tv) and torch.equal(boxes, tb) and torch.equal(labels, tl)
    assert torch.equal(rgt[valid], tg[tv])
    # image 2 has 41 candidates: nothing is dropped, so the sample is the whole candidate set, as the reference logic returns it
    _, rb, rl, _ = TF.label_and_sample(rh, prop[2, :40].cuda(), gts[2].cuda(), lambda k: torch.randperm(k))
    v = valid[2].cpu()
    assert int(v.sum()) == rb.shape[0] == 41
    got = sorted(map(tuple, boxes[2].cpu()[v].tolist()))
    assert got == sorted(map(tuple, rb.cpu().tolist())) and int((labels[2].cpu()[v] == 0).sum()) == int((rl == 0).sum())


def test_captured_step_recaptures_and_falls_back(oh):
    """GraphedTrainStep outside its comfort zone: more ground-truth boxes than the capacity -> a new capture with a doubled capacity;
    images of two sizes in one batch -> that iteration runs eagerly (no error recorded, the graph is kept); back to the captured shapes ->
    replay again."""
    from oracle import ref_train as T
    from detectron2.structures import Boxes, Instances
    from fewx.solver import GraphedTrainStep, build_optimizer
    shots = 4

    def item(seed, hw, n_gt):
        img, gt, sup, sbox = T.synth_train_inputs(seed, hw, n_gt=n_gt, shots=shots, support_hw=96)
        inst = Instances(hw)
        inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
        return {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}

    m, sd, cfg = _train_model(shots)
    opt = build_optimizer(cfg, m)
    opt.set_lr_factor(0.02)
    st = GraphedTrainStep(m, opt, warmup=2, gt_capacity=8)
    for i in range(4):                                      # 2 eager, capture at capacity 8, 2 replays
        losses = st([item(50 + i, (256, 320), 5)])
    assert st.error is None and st.eager_steps == 2 and st.replays == 2 and st.gt_capacity == 8
    g0 = st.graph
    losses = st([item(60, (256, 320), 11)])                 # 11 boxes > 8: capacity 16, a new graph
    assert st.error is None and st.gt_capacity == 16 and st.graph is not g0 and st.replays == 3
    assert all(bool(torch.isfinite(v)) for v in losses.values())
    g1 = st.graph
    losses = st([item(61, (256, 320), 4), item(62, (224, 320), 4)])      # two sizes: eager for this call
    assert st.error is None and st.eager_steps == 3 and st.graph is g1
    losses = st([item(63, (256, 320), 6)])
    assert st.replays == 4 and all(bool(torch.isfinite(v)) for v in losses.values())


def test_default_trainer_with_captured_step(oh, tmp_path):
    """trainer.graph_step = True: DefaultTrainer.run_step hands the iteration to fewx.solver.GraphedTrainStep -- three eager iterations,
    one capture, then replays on changing data (different images, different numbers of ground-truth boxes inside the capacity) under the
    warm-up LR schedule; the losses stay finite and are reported per iteration, the weights move, the checkpoint holds what the replays
    wrote."""
    import os
    from conftest import PKG
    from oracle import ref_train as T
    from detectron2.engine import DefaultTrainer
    from detectron2.structures import Boxes, Instances
    from fewx.config import get_cfg
    shots = 4

    def batches():
        i = 0
        while True:
            img, gt, sup, sbox = T.synth_train_inputs(40 + i % 5, (256, 320), n_gt=3 + i % 4, shots=shots, support_hw=96)
            inst = Instances((256, 320))
            inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
            yield [{"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}]
            i += 1

    class Trainer(DefaultTrainer):
        @classmethod
        def build_train_loader(cls, cfg):
            return batches()

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(PKG, "configs", "fsod", "finetune_vovnet.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "INPUT.FS.SUPPORT_SHOT", shots, "SOLVER.MAX_ITER", 9, "SOLVER.CHECKPOINT_PERIOD", 100,
                         "OUTPUT_DIR", str(tmp_path), "MODEL.WEIGHTS", ""])
    cfg.freeze()
    torch.manual_seed(0)
    tr = Trainer(cfg)
    tr.resume_or_load(resume=False)
    tr.graph_step = True
    seen = []
    orig = tr._write_metrics

    def spy(loss_dict, data_time, prefix=""):
        seen.append({k: float(v) for k, v in loss_dict.items()})
        return orig(loss_dict, data_time, prefix)

    tr._write_metrics = spy
    w0 = tr.model.conv3.weight.detach().clone()
    tr.train()
    g = tr._graphed
    assert g.error is None, g.error
    assert g.eager_steps == 3 and g.replays == 6 and tr.iter == 8
    assert len(seen) == 9 and all(all(v == v and abs(v) < 1e6 for v in d.values()) for d in seen)
    assert len({round(d["loss_centernet_loc"], 6) for d in seen}) >= 5                 # the replays report their own losses, not a stale buffer
    assert not torch.equal(tr.model.conv3.weight.detach(), w0)
    from detectron2.checkpoint import DetectionCheckpointer
    sd = torch.load(os.path.join(str(tmp_path), "model_final.pth"), map_location="cpu", weights_only=False)["model"]
    assert torch.equal(torch.as_tensor(sd["conv3.weight"]), tr.model.conv3.weight.detach().cpu())


def test_default_trainer_loop_with_synthetic_loader(oh, tmp_path):
    """The reference's training protocol end to end (ref:fsod_train_net.py:36-73,96-118): a DefaultTrainer subclass with a synthetic
    loader runs run_step + scheduler + periodic checkpoint, then resumes from the checkpoint (model, momentum, iteration)."""
    import os
    from conftest import PKG
    from oracle import ref_train as T
    from detectron2.engine import DefaultTrainer
    from detectron2.structures import Boxes, Instances
    from fewx.config import get_cfg
    shots = 4

    def batches():
        i = 0
        while True:
            img, gt, sup, sbox = T.synth_train_inputs(30 + i % 3, (256, 320), n_gt=5, shots=shots, support_hw=96)
            inst = Instances((256, 320))
            inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
            yield [{"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}]
            i += 1

    class Trainer(DefaultTrainer):
        @classmethod
        def build_train_loader(cls, cfg):
            return batches()

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(PKG, "configs", "fsod", "finetune_vovnet.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "INPUT.FS.SUPPORT_SHOT", shots, "SOLVER.MAX_ITER", 4, "SOLVER.CHECKPOINT_PERIOD", 2,
                         "OUTPUT_DIR", str(tmp_path), "MODEL.WEIGHTS", ""])
    cfg.freeze()
    torch.manual_seed(0)
    tr = Trainer(cfg)
    tr.resume_or_load(resume=False)
    w0 = tr.model.conv3.weight.detach().clone()
    tr.train()
    assert tr.iter == 3 and all(torch.isfinite(v) for v in tr.last_losses.values())
    assert set(tr.last_losses) == {"loss_cls_stage0", "loss_box_reg_stage0", "loss_centernet_loc", "loss_centernet_agn_pos", "loss_centernet_agn_neg"}
    assert not torch.equal(tr.model.conv3.weight.detach(), w0)
    g0 = tr.optimizer.param_groups[0]
    assert abs(g0["lr"] - g0["initial_lr"] * tr.scheduler.factor(4)) < 1e-12 and g0["initial_lr"] in (0.001, 0.002)
    files = sorted(os.listdir(tmp_path))
    assert "model_0000001.pth" in files and "model_0000003.pth" in files and "model_final.pth" in files and "last_checkpoint" in files
    w_end, mom_end = tr.model.conv3.weight.detach().clone(), tr.optimizer.bucket.momentum.clone()
    tr2 = Trainer(cfg)
    tr2.resume_or_load(resume=True)
    assert torch.equal(tr2.model.conv3.weight.detach(), w_end) and torch.equal(tr2.optimizer.bucket.momentum, mom_end)
    assert tr2.start_iter == 4


@pytest.mark.parametrize("tag", ["small", "full"])
def test_train_iteration_vs_reference_run(oh, golden, tag):
    """The product's training forward + backward against the EXECUTED reference (tests/golden/train_iter_ref_*.npz =
    ref:fewx/modeling/fsod/fsod_cen.py:151-308 run end to end, built by the reference's own __init__ from its logged config):
    `small` = the 5-shot configuration (SUPPORT_SHOT 4) on a 320x384 query, `full` = BASELINE configs[2]'s per-image shape
    (640x640, 24 support crops of 240x240).  Five losses, positive indices, train-mode proposals, and every parameter gradient with
    a PER-PARAMETER bound: gc/<name> in the fixture is the spread of the REFERENCE's own gradient on this sample under rounding-sized
    perturbations (its fp64 run and 10 fp32 runs with every parameter scaled by 1 + 1e-6 N(0,1), sampled ROIs held fixed).  The
    heat-map focal loss is full of hard decisions (ignore_high_fp, the 1e-4 cut, ReLU masks behind 13-20 dominant positive
    locations), so a few parameters move by 1e-2 under such perturbations -- e.g. the tower's GroupNorm bias by 1.5e-2 on the small
    sample, which is exactly what the CPU oracle shows between two host CPU models -- while most stay at 1e-5."""
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod.train_forward import train_forward
    from oracle import ref_train as T
    g = golden(f"train_iter_ref_{tag}")
    shots, hw = int(g["shots"]), tuple(int(v) for v in g["hw"])
    m, sd, cfg = _train_model(shots)
    img, gt, sup, sbox = T.synth_train_inputs(int(g["input_seed"]), hw, n_gt=int(g["n_gt"]), shots=shots, support_hw=int(g["support_hw"]))
    inst = Instances(hw)
    inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
    item = {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}
    over = {"boxes": torch.from_numpy(g["roi_boxes"]), "labels": torch.from_numpy(g["roi_labels"]), "gt": torch.from_numpy(g["roi_gt"])}
    losses, aux = train_forward(m, [item], return_aux=True, roi_override=over)
    n = len(g["pos_inds"])
    assert int(aux["pos_count"].item()) == n and np.array_equal(aux["pos_inds"][:n].cpu().numpy(), g["pos_inds"])
    pb, rb = aux["proposals"].cpu(), torch.from_numpy(g["proposals"])
    assert abs(len(pb) - len(rb)) <= 2, (len(pb), len(rb))
    d = (pb[:, None, :] - rb[None, :, :]).abs().amax(2).min(1)[0]       # order may swap on 1-ulp score ties: compare as sets
    assert float((d < 1e-2).float().mean()) >= 0.99, float((d < 1e-2).float().mean())
    # every row on which the two proposal sets differ is a near-tie of the train-mode NMS (IoU within 2e-5 of 0.9) or of the 2000th-score
    # cut, or the cascade of one (tests/near_tie.py: the greedy walk re-run on THIS path's candidates, the reference deciding near-ties
    # only).  Measured on MI355X, round 5: all 2000 rows match within 1e-2 px on both samples.
    from near_tie import guided_nms_explain
    det = aux["detect"]
    n_pre = int(det["counts"][0].item())
    ex = guided_nms_explain(det["pre_boxes"][:n_pre].cpu().numpy(), det["pre_scores"][:n_pre].cpu().numpy(), g["proposals"],
                            g["proposal_scores"], 0.9, post_topk=2000, box_tol=1e-2, score_rtol=2e-4)
    assert ex["unexplained"] == [] and ex["ambiguous"] <= 8, (ex["unexplained"][:6], ex["ambiguous"])
    for k in ("loss_cls_stage0", "loss_box_reg_stage0", "loss_centernet_loc", "loss_centernet_agn_pos", "loss_centernet_agn_neg"):
        want = float(g["loss/" + k])
        assert abs(float(losses[k].detach()) - want) <= 1e-4 * max(abs(want), 1e-3), (k, float(losses[k]), want)
    sum(losses.values()).backward()
    named = dict(m.named_parameters())
    report, n_checked = [], 0
    for key in g:
        if not key.startswith("gs/"):
            continue
        k = key[3:]
        f = named[k].grad.reshape(-1)
        smp = f[:: max(1, f.numel() // 1024)][:1024].cpu().numpy()
        err = float(np.abs(smp - g[key]).max()) / max(float(g["gn/" + k][1]), 1e-30)
        bound = max(2e-4, 3.0 * float(g["gc/" + k]))
        report.append((err / bound, err, bound, k))
        n_checked += 1
    report.sort()
    print("gradient error / bound (worst):", [(round(a, 3), f"{b:.2e}", f"{c:.2e}", d) for a, b, c, d in report[-6:]])
    assert n_checked == 73
    assert report[-1][0] <= 1.0, report[-4:]
    assert sorted(e[1] for e in report)[n_checked // 2] <= 2e-4
    for k in g["dead"]:
        p = named[str(k)]
        assert p.grad is None or float(p.grad.abs().max()) == 0.0, k


def test_train_forward_bs16_full_size(oh):
    """BASELINE configs[2] at its stated batch: 16 query images of 640x640 with 24 support crops each through ONE train_forward call.
    The five losses equal the mean of the 16 single-image calls run as 16 data-parallel ranks of the reference would run them
    (CenterNet normalisers = all-image totals / 16), every gradient is finite, and the flat gradient bucket has the size DESIGN 5 states (4,086,478 parameters with a gradient -> 16,365,568 bytes
    in 256-float chunks)."""
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod.train_forward import train_forward
    from fewx.solver.build import get_bucket
    from oracle import ref_train as T
    shots, B = 24, 16
    m, sd, cfg = _train_model(shots)
    items = []
    for b in range(B):
        img, gt, sup, sbox = T.synth_train_inputs(40 + b, (640, 640), n_gt=15 + b % 6, shots=shots, support_hw=240)
        inst = Instances((640, 640))
        inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
        items.append({"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()})
    torch.manual_seed(5)
    losses, aux = train_forward(m, items, return_aux=True)
    navg = (aux["cn_counts"] / B).clamp(min=1.0)                         # the reference's reduce_sum / num_gpus over B one-image ranks
    singles = []
    for b in range(B):                                                   # each image alone, on the ROIs the batched call sampled for it
        over = {"boxes": aux["roi_boxes"][b], "labels": aux["roi_labels"][b], "gt": aux["roi_gt"][b]}
        l1, _ = train_forward(m, [items[b]], return_aux=True, roi_override=over, cn_norm_avg=navg)
        singles.append({k: float(v.detach()) for k, v in l1.items()})
    for k, v in losses.items():
        want = sum(s[k] for s in singles) / B
        assert abs(float(v.detach()) - want) <= 2e-4 * max(abs(want), 1e-3), (k, float(v), want)
    # ONE image of the batch against the CPU oracle (everything above compares HIP with HIP): the oracle's single-image iteration on
    # the ROIs the batched call sampled for image 3, its raw CenterNet sums re-normalised by the batch's averaged normalisers
    b = 3
    leaf = {k: v.detach().clone() for k, v in sd.items()}
    over = {"boxes": aux["roi_boxes"][b].cpu(), "labels": aux["roi_labels"][b].cpu(), "gt": aux["roi_gt"][b].cpu()}
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    with torch.no_grad():
        ref = T.train_iteration(leaf, items[b]["image"], items[b]["instances"].gt_boxes.tensor.cpu(), items[b]["support_images"],
                                torch.from_numpy(items[b]["support_bboxes"]), lambda n: torch.randperm(n), roi_override=over)
    rs = R.centernet_losses(torch.cat([r.permute(0, 2, 3, 1).reshape(-1, 4) for r in ref["reg"]], 0),
                            torch.cat([h.permute(0, 2, 3, 1).reshape(-1) for h in ref["hm"]], 0), ref["pos_inds"], ref["reg_targets"],
                            ref["hm_targets"])["sums"]
    na = navg.cpu()
    want_b = {"loss_centernet_loc": float(rs[0] / na[0]), "loss_centernet_agn_pos": float(0.5 * 0.25 * -rs[2] / na[1]),
              "loss_centernet_agn_neg": float(0.5 * 0.75 * -rs[3] / na[1]),
              "loss_cls_stage0": float(ref["losses"]["loss_cls_stage0"]), "loss_box_reg_stage0": float(ref["losses"]["loss_box_reg_stage0"])}
    for k, want in want_b.items():
        assert abs(singles[b][k] - want) <= 5e-4 * max(abs(want), 1e-3), (k, singles[b][k], want)
    sum(losses.values()).backward()
    n_grad = 0
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert bool(torch.isfinite(p.grad).all()), k
            n_grad += p.numel()
    assert n_grad == 4086478
    bucket = get_bucket(m, cfg)
    assert bucket.grads.numel() * 4 == 16365568, bucket.grads.numel() * 4


def test_module_level_training_forwards(oh):
    """detectron2.layers.Conv2d and CenterNetHead follow the reference call protocol in training mode too (NCHW in/out, autograd)."""
    import torch.nn.functional as F
    from detectron2.layers import Conv2d
    g = torch.Generator().manual_seed(3)
    c = Conv2d(32, 48, kernel_size=3, padding=1, bias=True).cuda()
    x = torch.randn(2, 32, 9, 11, generator=g)
    xg = x.cuda().requires_grad_(True)
    y = c(xg)
    ref_x = x.clone().requires_grad_(True)
    w, b = c.weight.detach().cpu().requires_grad_(True), c.bias.detach().cpu().requires_grad_(True)
    ref = F.conv2d(ref_x, w, b, padding=1)
    _close(y, ref)
    up = torch.randn(ref.shape, generator=g)
    (y * up.cuda()).sum().backward(); (ref * up).sum().backward()
    _close(xg.grad, ref_x.grad); _close(c.weight.grad, w.grad); _close(c.bias.grad, b.grad)
    m, sd, cfg = _train_model(4)
    head = m.proposal_generator.centernet_head
    feats = [torch.randn(1, 128, 16 >> l, 24 >> l, generator=g) for l in range(3)]
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("proposal_generator.centernet_head.")}
    regs_ref, hms_ref = R.centernet_head(feats, {**sd, **leaf})
    _, regs, hms = head([f.cuda().requires_grad_(True) for f in feats])
    for l in range(3):
        _close(regs[l], regs_ref[l]); _close(hms[l], hms_ref[l])
    (sum((r ** 2).sum() for r in regs) + sum(h.sum() for h in hms)).backward()
    (sum((r ** 2).sum() for r in regs_ref) + sum(h.sum() for h in hms_ref)).backward()
    _close(head.bbox_tower[0].weight.grad, leaf["proposal_generator.centernet_head.bbox_tower.0.weight"].grad, tol=1e-4)
    _close(head.scales[1].scale.grad, leaf["proposal_generator.centernet_head.scales.1.scale"].grad, tol=1e-4)


def test_training_call_protocol_of_submodules(oh):
    """proposal_generator(images, features, gt_instances) -> (proposals, 3 losses) and roi_heads(images, features, support_box_features,
    proposals, targets) -> (proposals, 2 losses) in training mode, as the reference's detector calls them (fsod_cen.py:277-278)."""
    from detectron2.structures import Boxes, ImageList, Instances
    from detectron2.layers import nhwc_view
    from fewx.modeling.fsod.train_forward import head_train, proposal_losses_and_proposals
    m, sd, cfg = _train_model(4)
    g = torch.Generator().manual_seed(12)
    H, W = 128, 160
    gt = _rand_boxes(g, 5, W, H, lo=2.8, span=1.2)
    inst = Instances((H, W))
    inst.gt_boxes, inst.gt_classes = Boxes(gt.cuda()), torch.zeros(5, dtype=torch.int64).cuda()
    feats = {k: (torch.randn(1, 128, H >> s, W >> s, generator=g) * 0.5).cuda().requires_grad_(True) for k, s in (("p3", 3), ("p4", 4), ("p5", 5))}
    images = ImageList(torch.zeros(1, 3, H, W).cuda(), [(H, W)])
    pg, rh = m.proposal_generator, m.roi_heads
    props, l3 = pg(images, feats, [inst])
    assert set(l3) == {"loss_centernet_loc", "loss_centernet_agn_pos", "loss_centernet_agn_neg"} and len(props) == 1
    assert props[0].proposal_boxes.tensor.shape[1] == 4 and len(props[0].objectness_logits) == len(props[0].proposal_boxes.tensor)
    _, _, want, _ = proposal_losses_and_proposals(pg, head_train(pg.centernet_head, [nhwc_view(feats[k]) for k in ("p3", "p4", "p5")]), gt.cuda())
    for k in l3:
        assert abs(float(l3[k]) - float(want[k])) <= 1e-6 * max(abs(float(want[k])), 1e-3)
    sup8 = (torch.randn(4, 128, 8, 8, generator=g) * 0.3).cuda().requires_grad_(True)
    g1 = torch.Generator().manual_seed(5)
    out_props, l2 = rh(images, feats, [sup8, None], props, [inst], perm=lambda n: torch.randperm(n, generator=g1))
    assert set(l2) == {"loss_cls_stage0", "loss_box_reg_stage0"} and out_props is props
    (sum(l3.values()) + sum(l2.values())).backward()
    assert all(torch.isfinite(f.grad).all() and float(f.grad.abs().max()) > 0 for f in feats.values())
    assert float(sup8.grad.abs().max()) > 0 and float(rh.box_predictor[0].cls_score.weight.grad.abs().max()) > 0
