"""ORE_CONV_BF16 mode (include/ore_hip.h, BASELINE configs[4] "bf16 MFMA conv path + fp32 NMS") against the oracle's restatement of
the same mode (oracle/ref_model.py operand_precision("bf16"): operands rounded to bf16 where they enter a dense conv, everything
else fp32).  A product of two bf16 values is exact in fp32, so on the SAME inputs HIP and oracle differ by the fp32 summation order
only: single layers keep the fp32 tolerance (1e-4), and that is the parity statement of this mode.  Through the whole network the
two implementations decorrelate: a 1e-6 difference flips the bf16 rounding of ~1e-3 of the next layer's operands by a full bf16
ulp, which flips more roundings in the layer after, until both carry independent bf16 rounding noise (measured: rms 5e-3 between
HIP-bf16 and oracle-bf16, 2.5e-2 between either and fp32 -- tools/bf16_error_table.py, profiles/r01_bf16_error_table.txt).  The
whole-network test therefore bounds the distance to the bf16-mode oracle at that noise level and checks that it is several times
closer to it than to the fp32 oracle.  The reference has no reduced-precision path: this mode is pinned by construction (the fp32
oracle is the one pinned by reference-run fixtures)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import decode as odec
from oracle import ref_model as R

pytestmark = pytest.mark.gpu

TOL_LAYER = 1e-4       # one conv: summation order only
TOL_NET = 2.5e-2       # whole eval network, max-norm, vs the bf16-mode oracle: independent bf16 rounding noise (see above)


@pytest.fixture(scope="module")
def ore():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import orehip
    orehip.lib()
    return orehip


@pytest.fixture()
def bf16(ore):
    prev = ore.set_conv_precision("bf16")
    yield
    ore.set_conv_precision(prev)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(x):
    return x.permute(0, 3, 1, 2).cpu()


def test_precision_switch_round_trips(ore):
    assert ore.get_conv_precision() == "fp32"
    assert ore.set_conv_precision("bf16") == "fp32" and ore.get_conv_precision() == "bf16"
    assert ore.set_conv_precision("fp32") == "bf16" and ore.get_conv_precision() == "fp32"
    with pytest.raises(KeyError):
        ore.set_conv_precision("fp8")


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,splitk", [
    (1, 20, 20, 384, 112, 3, 1, 0),    # small-M tile, K split across the block's waves (+ cross-block split-K)
    (2, 17, 23, 352, 256, 1, 1, 0),    # 1x1 concat, odd spatial
    (1, 9, 7, 16, 5, 3, 1, 0),         # Cout = 5 (head), padded to 16
    (1, 80, 80, 128, 128, 3, 1, 0),    # 3x3 patch kernel
    (3, 33, 31, 64, 80, 3, 2, 0),      # stride 2
    (1, 160, 160, 64, 64, 3, 1, 0),    # large-M tiles
    (1, 320, 320, 64, 64, 3, 1, 0),    # weight-stationary kernel (stem_2 shape)
    (1, 160, 160, 128, 64, 3, 1, 0),   # stage-2 layer 0 shape
])
def test_conv_bf16_operands_vs_oracle(ore, bf16, B, H, W, Cin, Cout, k, stride, splitk):
    g = torch.Generator().manual_seed(B * 1000 + H + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    sc = torch.rand(Cout, generator=g) + 0.5
    sh = torch.randn(Cout, generator=g) * 0.1
    with R.operand_precision("bf16"):
        ref = F.relu(R.dense_conv(x, w, None, stride, k // 2) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    full = F.relu(F.conv2d(x, w, None, stride, k // 2) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    y = ore.conv2d(nhwc(x), ore.pack_conv_weight(w).cuda(), Cout, k, stride, scale=sc.cuda(), shift=sh.cuda(), relu_cout=Cout, splitk=splitk)
    got = nchw(y).numpy()
    assert rel_err(got, ref.numpy()) < TOL_LAYER
    assert 1e-4 < rel_err(got, full.numpy()) < 2e-2          # it really is the reduced-precision path, and a sane one


@pytest.mark.parametrize("bm,bn,H,W,Cin,Cout,k,stride", [(64, 128, 80, 80, 352, 256, 1, 1), (64, 64, 80, 80, 256, 128, 1, 1), (32, 64, 83, 79, 256, 112, 1, 1),
                                                     (64, 128, 160, 160, 64, 128, 3, 2)])
def test_conv_gd_bf16_operand_builds_vs_oracle(ore, bf16, bm, bn, H, W, Cin, Cout, k, stride):
    """The bf16-OPERAND builds of k_conv_gd (round 5: fragments rounded as they leave LDS, one v_mfma_f32_16x16x16_bf16 per tile), each
    forced on a layer it can serve, against the oracle's restatement of the mode at the fp32 tolerance (a bf16 x bf16 product is
    exact in fp32), incl. a partial last row tile, a Cout that is not a multiple of the tile and the 3x3 stride-2 chunk table."""
    g = torch.Generator().manual_seed(bm + bn + H + Cin)
    x = torch.randn(1, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    with R.operand_precision("bf16"):
        ref = F.relu(R.dense_conv(x, w, None, stride, k // 2) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    L = ore.lib()
    assert L.ore_conv_set_plan_override(-15, bm, bn, 4, 0) == 0
    try:
        y = ore.conv2d(nhwc(x), ore.pack_conv_weight(w).cuda(), Cout, k, stride, scale=sc.cuda(), shift=sh.cuda(), relu_cout=Cout)
    finally:
        L.ore_conv_set_plan_override(-15, 0, 0, 0, 0)
    got = nchw(y).numpy()
    assert rel_err(got, ref.numpy()) < TOL_LAYER
    full = F.relu(F.conv2d(x, w, None, stride, k // 2) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    assert 1e-4 < rel_err(got, full.numpy()) < 2e-2


def test_conv_bf16_input_affine_rounds_after_the_affine(ore, bf16):
    """GN / eSE folds: relu(x * mul + add) is computed in fp32 while the tile is staged, THEN rounded as the MFMA operand."""
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 11, 14
    buf = torch.randn(B, 96, H, W, generator=g)
    w = torch.randn(32, 48, 3, 3, generator=g) * 0.05
    mul = torch.rand(B, 48, generator=g) + 0.5
    add_in = torch.randn(B, 48, generator=g) * 0.2
    bias = torch.randn(32, generator=g)
    xin = F.relu(buf[:, 32:80] * mul[:, :, None, None] + add_in[:, :, None, None])
    with R.operand_precision("bf16"):
        ref = R.dense_conv(xin, w, bias, 1, 1)
    out = ore.conv2d(nhwc(buf), ore.pack_conv_weight(w).cuda(), 32, 3, 1, in_coff=32, Cin=48, shift=bias.cuda(), in_mul=mul.cuda(),
                     in_add=add_in.cuda(), in_relu=True)
    assert rel_err(nchw(out).numpy(), ref.numpy()) < TOL_LAYER


def test_engine_bf16_vs_oracle_bf16(ore):
    """Whole eval hot path at the BASELINE shape in bf16 mode: feature maps against the bf16-mode oracle; the detection tail is fp32
    and stays BIT-EXACT against ref_decode.c on the same head outputs; the engine keeps its mode after the global flag is reset."""
    sd = R.synth_state_dict(0)
    prev = ore.set_conv_precision("bf16")
    try:
        e = ore.Engine(max_batch=1, max_h=640, max_w=640)
    finally:
        ore.set_conv_precision(prev)
    assert ore.get_conv_precision() == "fp32"
    e.load_state_dict(sd)
    e.set_support(R.synth_support(0))
    e.finalize()
    img = R.synth_image(0)
    with R.operand_precision("bf16"):
        ref = R.eval_dense(img, sd, R.synth_support(0))
    ref32 = R.eval_dense(img, sd, R.synth_support(0))
    for use_graph in (False, True, True):
        e.eval_forward(img.cuda(), use_graph=use_graph)
        torch.cuda.synchronize()
        for l, k in enumerate(("p3", "p4", "p5")):
            s = 640 >> (l + 3)
            got = e.buffer(k, (1, s, s)).cpu().numpy()
            assert rel_err(got, ref["features"][k].numpy()) < TOL_NET, k
            assert 1e-4 < rel_err(got, ref32["features"][k].numpy()) < 8e-2, k
            rms = lambda a, b: float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))      # noqa: E731
            assert 2.5 * rms(got, ref["features"][k].numpy()) < rms(got, ref32["features"][k].numpy()), k
            assert rel_err(e.buffer(f"pos{l + 3}", (1, s, s)).cpu().numpy(), ref["pos_features"][l].numpy()) < TOL_NET
            hd = e.buffer(f"head{l + 3}", (1, s, s)).cpu()
            assert rel_err(hd[:, :4].numpy(), ref["reg"][l].numpy()) < TOL_NET
            assert rel_err(hd[:, 4:5].numpy(), ref["hm"][l].numpy()) < TOL_NET
        hms, regs = [], []
        for l in range(3):
            s = 640 >> (l + 3)
            hd = e.buffer(f"head{l + 3}").cpu().numpy().reshape(s, s, 5)
            hms.append(np.ascontiguousarray(hd[..., 4]))
            regs.append(np.ascontiguousarray(hd[..., :4]))
        want = odec.decode_nms(hms, regs, (8, 16, 32), 1e-5, 1000, 0.6, 256)
        boxes, scores, keep = e.proposals()
        assert np.array_equal(keep.cpu().numpy(), want["keep"]) and len(want["keep"]) > 0
        assert np.array_equal(boxes.cpu().numpy(), want["boxes"]) and np.array_equal(scores.cpu().numpy(), want["scores"])
    e.close()


# ---------------------------------------------------------------------------------------------------------------------------
# training form of the mode (BASELINE configs[4] names 5-shot *training*): forward, data gradient and weight gradient of every layer
# that runs on the MFMA conv kernel round their operands to bf16; the oracle restates that (operand_precision("bf16", train=True)).
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,relu", [
    (2, 20, 24, 128, 128, 3, True),     # FPN output / head tower shape class: k_wgrad3 + bf16 data-gradient conv
    (1, 10, 12, 512, 128, 1, False),    # lateral 1x1: k_wgrad
    (1, 1, 96, 1024, 128, 1, True),     # a Linear over rows (fc1 / SM_Block class)
    (3, 9, 7, 128, 16, 3, False),       # (reg | hm) head conv: Cout 5 padded to 16 is handled by the caller; here Cout = 16
])
def test_conv_backward_bf16_vs_oracle(ore, bf16, B, H, W, Cin, Cout, k, relu):
    """One conv layer forward + backward in the bf16-operand mode against the oracle's restatement on the SAME inputs: a product of
    two bf16 values is exact in fp32 and both sides accumulate in fp32, so output, data gradient, weight gradient and bias gradient
    differ by the summation order only -- the fp32 tolerance (1e-4) is asserted."""
    from orehip import autograd as A
    g = torch.Generator().manual_seed(5 + Cin + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    up = torch.randn(B, Cout, H, W, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    with R.operand_precision("bf16", train=True):
        y = R.dense_conv(xr, wr, br, 1, k // 2)
        y = F.relu(y) if relu else y
        (y * up).sum().backward()
    xg, wg, bg = nhwc(x).requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yh = A.conv(xg, wg, bg, None, None, relu)
    (yh * nhwc(up)).sum().backward()
    assert rel_err(nchw(yh.detach()).numpy(), y.detach().numpy()) < TOL_LAYER
    assert rel_err(nchw(xg.grad).numpy(), xr.grad.numpy()) < TOL_LAYER
    assert rel_err(wg.grad.cpu().numpy(), wr.grad.numpy()) < TOL_LAYER
    assert rel_err(bg.grad.cpu().numpy(), br.grad.numpy()) < TOL_LAYER
    # and the mode is really on: the fp32 result of the same layer is a bf16 rounding error away
    y32 = F.conv2d(x, w, b, 1, k // 2)
    y32 = F.relu(y32) if relu else y32
    assert rel_err(nchw(yh.detach()).numpy(), y32.numpy()) > 5e-4


def test_train_iteration_5shot_bf16_vs_oracle(ore, bf16):
    """BASELINE configs[4]'s training form: one 5-shot (SUPPORT_SHOT 4) iteration -- forward, backward through the HIP data- and
    weight-gradient kernels -- in the bf16-operand mode against the oracle's restatement of the same mode on the same sample and the
    same sampled ROIs.  Through 40 layers forward and back the two implementations carry independent bf16 rounding noise (module
    docstring), so this is the loose whole-iteration class: losses within 3 %, positive indices exact (fp32 targets), the live
    parameters' gradients within the bf16 noise level of the bf16 oracle (median 5 %, 90th percentile 20 % of the tensor's max) AND
    closer to it than to the fp32 oracle, the dead parameters dead; the tight statement of the mode is
    test_conv_backward_bf16_vs_oracle (1e-4 per layer, forward and both gradients)."""
    import os
    from conftest import PKG
    from fewx.config import get_cfg
    from detectron2.modeling import build_model
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod.train_forward import train_forward
    from oracle import ref_train as T
    shots = 4
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(PKG, "configs", "fsod", "finetune_vovnet.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "INPUT.FS.SUPPORT_SHOT", shots])
    cfg.freeze()
    torch.manual_seed(0)
    m = build_model(cfg)
    sd = R.synth_roi_state(R.synth_state_dict(0), 0)
    sd["roi_heads.box_head.0.fc1.weight"] = sd["roi_heads.box_head.0.fc1.weight"] * 0.02
    sd["proposal_generator.centernet_head.agn_hm.bias"] = torch.full((1,), -2.0)
    m.load_state_dict(sd, strict=False)
    m.train()
    for lvl in (3, 4, 5):
        getattr(m, f"vip_p{lvl}").reweighting.drop.p = 0.0
    img, gt, sup, sbox = T.synth_train_inputs(0, (320, 384), n_gt=9, shots=shots, support_hw=112)
    leaf = T.leaf_state(sd)
    gen = torch.Generator().manual_seed(11)
    with R.operand_precision("bf16", train=True):
        ref = T.train_iteration(leaf, img, gt, sup, sbox, lambda n: torch.randperm(n, generator=gen))
        sum(ref["losses"].values()).backward()
    inst = Instances((320, 384))
    inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
    item = {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}
    over = {"boxes": ref["roi_boxes"], "labels": ref["roi_labels"], "gt": ref["roi_gt"]}
    losses, aux = train_forward(m, [item], return_aux=True, roi_override=over)
    assert int(aux["pos_count"].item()) == len(ref["pos_inds"]) and torch.equal(aux["pos_inds"][: len(ref["pos_inds"])].cpu(), ref["pos_inds"])
    for k, v in ref["losses"].items():
        assert abs(float(losses[k].detach()) - float(v.detach())) <= 3e-2 * max(abs(float(v.detach())), 1e-3), (k, float(losses[k]), float(v))
    sum(losses.values()).backward()
    named = dict(m.named_parameters())
    errs = []
    for k, t in leaf.items():
        if not t.requires_grad:
            continue
        p = named[k]
        if t.grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
        errs.append((float((p.grad.cpu() - t.grad).abs().max()) / max(float(t.grad.abs().max()), 1e-8), k))
    errs.sort()
    assert len(errs) == 73
    assert errs[len(errs) // 2][0] <= 5e-2 and errs[int(len(errs) * 0.9)][0] <= 2e-1, (errs[len(errs) // 2], errs[-6:])
    # ... and the gradients are those of the bf16 mode, not of the fp32 path: closer to the bf16 oracle than to the fp32 oracle
    leaf32 = T.leaf_state(sd)
    ref32 = T.train_iteration(leaf32, img, gt, sup, sbox, lambda n: torch.randperm(n), roi_override=over)
    sum(ref32["losses"].values()).backward()
    e32 = sorted(float((named[k].grad.cpu() - t.grad).abs().max()) / max(float(t.grad.abs().max()), 1e-8)
                 for k, t in leaf32.items() if t.requires_grad and t.grad is not None)
    assert errs[len(errs) // 2][0] < 0.7 * e32[len(e32) // 2], (errs[len(errs) // 2][0], e32[len(e32) // 2])


# ---- bf16 STORAGE mode (BASELINE configs[4] as a byte-saving path): bf16 tensors in HBM and LDS, v_mfma_f32_16x16x32_bf16 ---------
def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride", [
    (1, 20, 20, 384, 112, 3, 1),    # deep K, Cout not a multiple of 32
    (1, 20, 20, 112, 112, 3, 1),    # Cin % 32 == 16: the last K chunk of a tap reads 16 channels further, against zero weights
    (1, 40, 40, 544, 384, 1, 1),    # 1x1 concat
    (1, 80, 80, 80, 80, 3, 1),      # k_conv_kw at M = 6400, Cin = 80
    (1, 80, 80, 352, 256, 1, 1),    # k_conv_gs 64x64 tiles
    (2, 17, 23, 352, 256, 1, 1),    # odd sizes, two images
    (1, 9, 7, 16, 5, 3, 1),         # tiny, Cout = 5
    (3, 33, 31, 64, 80, 3, 2),      # stride 2
    (1, 96, 96, 64, 64, 3, 1),      # k_conv_gs on a 3x3 layer (M = 9216)
    (1, 80, 80, 128, 128, 3, 1),    # FPN output3 / tower class: K-split weight-stationary kernel (8 waves)
    (1, 160, 160, 64, 128, 3, 2),   # stem_3 class: weight-stationary stride-2 kernel
    (3, 111, 93, 64, 64, 3, 2),     # the same on odd sizes
    (2, 70, 50, 128, 64, 3, 1),     # stage-2 layer 0 class on a size with partial tiles
])
def test_conv_bf16_storage_vs_oracle(B, H, W, Cin, Cout, k, stride):
    """bf16 input / weight / output tensors.  A product of two bf16 numbers is exact in fp32 and the accumulation is fp32, so against
    F.conv2d on the SAME bf16 values (upcast) the pre-rounding result differs by summation order only; the stored output may then
    differ by one bf16 ulp where that difference straddles a rounding boundary."""
    import orehip as ore
    g = torch.Generator().manual_seed(B * 1000 + H + Cin + Cout + k)
    x = _bf(torch.randn(B, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    sc = torch.rand(Cout, generator=g) + 0.5
    sh = torch.randn(Cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(x.float(), w.float(), None, stride, k // 2) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    xh = torch.zeros(B * H * W + 1, Cin, dtype=torch.bfloat16)                      # one spare row: the Cin % 32 == 16 over-read
    xh[:-1] = x.permute(0, 2, 3, 1).reshape(-1, Cin)
    xd = xh.cuda()[:-1].view(B, H, W, Cin)
    wp = ore.pack_conv_weight_bf16(w.float()).cuda()
    y32 = ore.conv2d(xd, wp, Cout, k, stride, scale=sc.cuda(), shift=sh.cuda(), relu_cout=Cout, out_f32=True)
    got32 = y32.cpu().permute(0, 3, 1, 2)
    assert float((got32 - ref).abs().max() / ref.abs().max()) < 1e-4             # fp32 output: the fp32 tolerance
    y = ore.conv2d(xd, wp, Cout, k, stride, scale=sc.cuda(), shift=sh.cuda(), relu_cout=Cout)
    assert y.dtype == torch.bfloat16
    # the bf16 output is the fp32 result rounded once (another kernel may serve the bf16-output launch -- the weight-stationary 3x3
    # kernel at M >= 6000 --, so a different summation order can move a value across a rounding boundary: one ulp)
    assert _ulp_close(y.cpu(), y32.cpu())
    y2 = ore.conv2d(xd, wp, Cout, k, stride, scale=sc.cuda(), shift=sh.cuda(), relu_cout=Cout)
    assert torch.equal(y, y2)


KD_BUILDS = [(16, 16, 4, 4), (16, 16, 4, 8), (16, 16, 8, 4), (16, 16, 16, 2), (16, 32, 4, 4), (16, 32, 8, 2), (32, 32, 4, 4), (32, 32, 8, 2),
             (16, 48, 4, 4), (16, 48, 8, 2), (16, 64, 4, 2), (16, 80, 4, 2), (32, 64, 4, 2), (32, 80, 4, 2), (64, 64, 4, 2), (32, 48, 4, 2),
             (32, 16, 4, 4)]


@pytest.mark.gpu
@pytest.mark.parametrize("bm,bn,nw,sb", KD_BUILDS)
@pytest.mark.parametrize("H,W,Cin,Cout,k", [(20, 20, 112, 112, 3),     # Cin % 32 == 16: half-empty last chunk of every tap
                                            (13, 19, 544, 384, 1),     # 1x1, odd size, Cout wider than every tile
                                            (9, 11, 96, 40, 3)])       # tiny, Cout = 40
def test_conv_kd_bf16_storage_every_build(bm, bn, nw, sb, H, W, Cin, Cout, k):
    """k_conv_kd's bf16-STORAGE builds (round 4: the staging is byte for byte the fp32 one, a 64-byte row is 32 channels, one
    v_mfma_f32_16x16x32_bf16 per step): every (tile, waves, batch) build against F.conv2d on the same bf16 values -- fp32 output at the
    fp32 tolerance, bf16 output within one ulp of it, column sums of the STORED values, bit-reproducible."""
    import orehip as ore
    if bn > (Cout + 15) // 16 * 16:
        pytest.skip("tile wider than the layer: the launcher narrows it to another build")
    g = torch.Generator().manual_seed(H + Cin + Cout + k + bm + bn + nw)
    x = _bf(torch.randn(1, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    sh = torch.randn(Cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(x.float(), w.float(), sh, 1, k // 2))
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = ore.pack_conv_weight_bf16(w.float()).cuda()
    L = ore.lib()
    L.ore_conv_set_plan_override(-13, bm, bn, nw, sb)
    try:
        y32 = ore.conv2d(xd, wp, Cout, k, 1, shift=sh.cuda(), relu_cout=Cout, out_f32=True)
        y, cs = ore.conv2d(xd, wp, Cout, k, 1, shift=sh.cuda(), relu_cout=Cout, want_colsum=True)
        y2 = ore.conv2d(xd, wp, Cout, k, 1, shift=sh.cuda(), relu_cout=Cout)
    finally:
        L.ore_conv_set_plan_override(-13, 0, 0, 0, 0)
    assert float((y32.cpu().permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max()) < 1e-4
    assert y.dtype == torch.bfloat16 and _ulp_close(y.cpu(), y32.cpu()) and torch.equal(y, y2)
    want_cs = y.float().sum((0, 1, 2)).cpu()
    assert float((cs.sum(0)[:Cout].cpu() - want_cs).abs().max()) <= 1e-5 * float(want_cs.abs().max() + y.float().abs().sum().cpu() / Cout * 1e-2)


@pytest.mark.gpu
def test_conv_bf16_storage_slices_add_colsum_levels():
    """Channel-slice input / output inside wider bf16 buffers, the FPN top-down add from a bf16 tensor, fused column sums of the ROUNDED
    output, three pyramid levels in one launch with per-level epilogue parameters and fp32 output (the head's last conv)."""
    import orehip as ore
    g = torch.Generator().manual_seed(21)
    B, H, W = 1, 20, 24
    buf = _bf(torch.randn(B, 160, H, W, generator=g))                              # read channels 32..143 (112)
    w = _bf(torch.randn(80, 112, 3, 3, generator=g) * 0.03)
    sh = torch.randn(80, generator=g) * 0.1
    out = torch.full((B, H, W, 128), 7.0, dtype=torch.bfloat16).cuda()
    xin = buf.permute(0, 2, 3, 1).contiguous().cuda()
    wp = ore.pack_conv_weight_bf16(w.float()).cuda()
    y, cs = ore.conv2d(xin, wp, 80, 3, 1, in_coff=32, Cin=112, shift=sh.cuda(), relu_cout=80, out=out, out_coff=16, want_colsum=True)
    ref = F.relu(F.conv2d(buf[:, 32:144].float(), w.float(), sh, 1, 1))
    got = out[..., 16:96].float().cpu().permute(0, 3, 1, 2)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-2                # one bf16 rounding of the output
    assert float(out[..., :16].float().min()) == 7.0 and float(out[..., 96:].float().max()) == 7.0
    want_cs = out[..., 16:96].float().sum((0, 1, 2)).cpu()                         # column sums of what was STORED
    assert float((cs.sum(0)[:80].cpu() - want_cs).abs().max() / want_cs.abs().max()) < 1e-5
    # 1x1 lateral with the nearest-2x top-down add
    top = _bf(torch.randn(B, 128, (H + 1) // 2, (W + 1) // 2, generator=g))
    x2 = _bf(torch.randn(B, 256, H, W, generator=g))
    w2 = _bf(torch.randn(128, 256, 1, 1, generator=g) * 0.05)
    b2 = torch.randn(128, generator=g) * 0.1
    y2 = ore.conv2d(x2.permute(0, 2, 3, 1).contiguous().cuda(), ore.pack_conv_weight_bf16(w2.float()).cuda(), 128, 1, 1, shift=b2.cuda(),
                    add=top.permute(0, 2, 3, 1).contiguous().cuda(), out_f32=True)
    ref2 = F.conv2d(x2.float(), w2.float(), b2) + F.interpolate(top.float(), scale_factor=2, mode="nearest")[:, :, :H, :W]
    assert float((y2.cpu().permute(0, 3, 1, 2) - ref2).abs().max() / ref2.abs().max()) < 1e-4
    # three levels, per-level scale / shift, Cout = 5, fp32 out
    HW = [(12, 16), (6, 8), (3, 4)]
    xs = [_bf(torch.randn(1, 128, h, w_, generator=g)) for h, w_ in HW]
    w5 = _bf(torch.randn(5, 128, 3, 3, generator=g) * 0.05)
    sc3, sh3 = torch.rand(3, 16, generator=g) + 0.5, torch.randn(3, 16, generator=g) * 0.1
    rows = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, 128) for t in xs] + [torch.zeros(1, 128, dtype=torch.bfloat16)], 0).contiguous().cuda()
    nrow = rows.shape[0] - 1
    outh = torch.zeros(nrow, 8, device="cuda")
    ore.conv2d_levels(rows[:nrow], HW, 1, ore.pack_conv_weight_bf16(w5.float()).cuda(), 5, 3, scale=sc3.cuda().contiguous(),
                      shift=sh3.cuda().contiguous(), ep_stride=16, relu_cout=4, out=outh, out_f32=True)
    r0 = 0
    for l, ((h, w_), t) in enumerate(zip(HW, xs)):
        r = F.conv2d(t.float(), w5.float(), None, 1, 1) * sc3[l, :5].view(1, -1, 1, 1) + sh3[l, :5].view(1, -1, 1, 1)
        r[:, :4] = F.relu(r[:, :4])
        want = r[0].permute(1, 2, 0).reshape(-1, 5)
        assert float((outh[r0:r0 + h * w_, :5].cpu() - want).abs().max() / want.abs().max()) < 1e-4
        r0 += h * w_


@pytest.mark.gpu
@pytest.mark.parametrize("HW,B", [([(80, 80), (40, 40), (20, 20)], 1), ([(12, 20), (6, 10), (3, 5)], 2)])
def test_conv_bf16_storage_per_level_weights(HW, B):
    """bf16 storage twin of test_conv_winograd_per_level_weights: three 3x3 128 -> 128 layers over three pyramid levels in ONE launch
    of the weight-stationary kernel (ore_conv_desc.w_level_stride: every block loads its level's weights), bf16 in, bf16 out."""
    import orehip as ore
    g = torch.Generator().manual_seed(HW[0][0] + B)
    Cc = 128
    xs = [_bf(torch.randn(B, Cc, h, w_, generator=g)) for h, w_ in HW]
    ws = [_bf(torch.randn(Cc, Cc, 3, 3, generator=g) * 0.03) for _ in HW]
    bs = [torch.randn(Cc, generator=g) * 0.1 for _ in HW]
    rows = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, Cc) for t in xs] + [torch.zeros(1, Cc, dtype=torch.bfloat16)], 0).contiguous().cuda()
    nrow = rows.shape[0] - 1
    wp = torch.stack([ore.pack_conv_weight_bf16(w.float()) for w in ws]).contiguous().cuda()
    out = torch.full((nrow, 2 * Cc), 7.0, dtype=torch.bfloat16).cuda()
    ore.conv2d_levels(rows[:nrow], HW, B, wp, Cc, 3, shift=torch.stack(bs).contiguous().cuda(), ep_stride=Cc, out=out, out_coff=Cc,
                      w_level_stride=wp.shape[1])
    assert float(out[:, :Cc].float().min()) == 7.0 and float(out[:, :Cc].float().max()) == 7.0
    r0 = 0
    for (h, w_), t, w, b in zip(HW, xs, ws, bs):
        ref_l = F.conv2d(t.float(), w.float(), b, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cc)
        got = out[r0:r0 + B * h * w_, Cc:].float().cpu()
        assert float((got - ref_l).abs().max() / ref_l.abs().max()) < 1e-2           # one bf16 rounding of the output
        r0 += B * h * w_
    with pytest.raises(ore.OreError):                                                # 1x1 layers have no per-level form
        ore.conv2d_levels(rows[:nrow], HW, B, ore.pack_conv_weight_bf16(torch.randn(Cc, Cc, 1, 1)).cuda(), Cc, 1, w_level_stride=wp.shape[1])


def _bf16s_engine(ore, sd, hw, roi=False):
    prev = ore.set_conv_precision("bf16s")
    try:
        e = ore.Engine(max_batch=1, max_h=hw[0], max_w=hw[1])
    finally:
        ore.set_conv_precision(prev)
    assert ore.get_conv_precision() == "fp32"
    e.load_state_dict(sd)
    e.set_support(R.synth_support(0))
    e.finalize()
    return e


def _ulp_close(got, want, ulps=1.0):
    """got (bf16 tensor stored by the engine) vs want (fp32 reference BEFORE rounding): within `ulps` bf16 ulps of the reference value
    (half an ulp is the rounding itself; the rest covers a summation-order difference straddling a rounding boundary)."""
    got, want = got.float(), want.float()
    ulp = torch.maximum(want.abs(), torch.tensor(1e-30)) * 2.0 ** -8 + 1e-6 * float(want.abs().max())
    return bool(((got - want).abs() <= ulps * ulp).all())


@pytest.mark.gpu
def test_engine_bf16_storage_every_kernel_exact_on_its_own_inputs(ore):
    """ORE_CONV_BF16S engine, layer by layer: each kernel's bf16 OUTPUT buffer against the oracle's fp32 arithmetic applied to that
    kernel's own bf16 INPUT buffer(s) -- one bf16 rounding (plus summation order) apart, whatever noise the layers before accumulated."""
    sd = R.synth_state_dict(0)
    H, W = 160, 192
    e = _bf16s_engine(ore, sd, (H, W))
    img = R.synth_image(5, H, W)
    e.eval_forward(img.cuda(), use_graph=False)
    torch.cuda.synchronize()
    B = lambda n, h, w: e.buffer(n, (1, h, w)).cpu()                                   # noqa: E731
    bu = "backbone.bottom_up."
    s1, s2, s3 = B("stem1", H // 2, W // 2), B("stem2", H // 2, W // 2), B("stem3", H // 4, W // 4)
    assert s1.dtype == torch.bfloat16 and e.buffer("head3").dtype == torch.float32
    assert _ulp_close(s1, R.conv_bn_relu(R.preprocess(img), sd, bu + "stem.stem_1", 2, 1))
    with R.operand_precision("bf16"):                                                # rounds the weights; the inputs are bf16 already
        assert _ulp_close(s2, R.conv_bn_relu(s1.float(), sd, bu + "stem.stem_2", 1, 1))
        assert _ulp_close(s3, R.conv_bn_relu(s2.float(), sd, bu + "stem.stem_3", 2, 1))
        # stage 3: max-pool of (stage-2 output x gate), three 3x3 layers writing slices of the concat buffer, the 1x1 concat conv
        st2, g2 = B("stage2", H // 4, W // 4).float(), e.buffer("gate2").cpu().view(1, -1, 1, 1)
        cat3 = B("cat3", H // 8, W // 8)
        pooled = F.max_pool2d(st2 * g2, 3, 2, ceil_mode=True)
        assert _ulp_close(cat3[:, :112], pooled)
        mod, pre = "OSA3_1", bu + "stage3.OSA3_1."
        y = cat3[:, :112].float()
        for i in range(3):
            y_ref = R.conv_bn_relu(y, sd, f"{pre}layers.{i}.{mod}_{i}", 1, 1)
            got = cat3[:, 112 + 80 * i: 192 + 80 * i]
            assert _ulp_close(got, y_ref), i
            y = got.float()
        st3 = B("stage3", H // 8, W // 8)
        assert _ulp_close(st3, R.conv_bn_relu(cat3.float(), sd, f"{pre}concat.{mod}_concat", 1, 0))
        # the eSE gate: fp32, from the mean of the STORED (rounded) stage output
        gate_ref = F.relu6(F.conv2d(st3.float().mean((2, 3), keepdim=True), sd[pre + "ese.fc.weight"], sd[pre + "ese.fc.bias"]) + 3.0) / 6.0
        assert float((e.buffer("gate3").cpu().view(-1) - gate_ref.view(-1)).abs().max()) < 1e-5
        # FPN level 5: lateral on round(W * g), output conv; level 4: lateral + nearest-2x add of lat5
        st5, g5 = B("stage5", H // 32, W // 32).float(), e.buffer("gate5").cpu().view(1, -1, 1, 1)
        lat5 = B("lat5", H // 32, W // 32)
        w5 = (sd["backbone.fpn_lateral5.weight"] * g5).bfloat16().float()
        assert _ulp_close(lat5, F.conv2d(st5, w5, sd["backbone.fpn_lateral5.bias"]))
        p5 = B("p5", H // 32, W // 32)
        assert _ulp_close(p5, R.dense_conv(lat5.float(), sd["backbone.fpn_output5.weight"], sd["backbone.fpn_output5.bias"], padding=1))
        st4, g4 = B("stage4", H // 16, W // 16).float(), e.buffer("gate4").cpu().view(1, -1, 1, 1)
        w4 = (sd["backbone.fpn_lateral4.weight"] * g4).bfloat16().float()
        lat4_ref = F.conv2d(st4, w4, sd["backbone.fpn_lateral4.bias"]) + F.interpolate(lat5.float(), scale_factor=2.0, mode="nearest")
        assert _ulp_close(B("lat4", H // 16, W // 16), lat4_ref)
    # correlation (fp32 arithmetic on the bf16 query), conv3, tower, GroupNorm + ReLU + (reg | hm) with fp32 outputs
    sup = R.synth_support(0)
    for l, k in enumerate(("p3", "p4", "p5")):
        h, w = H >> (l + 3), W >> (l + 3)
        q = B(k, h, w).float()
        C = q.shape[1]
        k11 = F.adaptive_avg_pool2d(sup[k], (1, 1)).permute(1, 0, 2, 3)
        k13 = F.adaptive_avg_pool2d(sup[k], (1, 3)).permute(1, 0, 2, 3)
        k31 = F.adaptive_avg_pool2d(sup[k], (3, 1)).permute(1, 0, 2, 3)
        a = F.relu(F.conv2d(F.relu(F.conv2d(q, k11, groups=C)), k11, groups=C))
        b = F.relu(F.conv2d(F.relu(F.conv2d(q, k13, padding=(0, 1), groups=C)), k31, padding=(1, 0), groups=C))
        attn = B(f"attn{l + 3}", h, w)
        assert _ulp_close(attn, a + b + q), k
        with R.operand_precision("bf16"):
            pos = B(f"pos{l + 3}", h, w)
            assert _ulp_close(pos, F.relu(R.dense_conv(torch.cat((attn.float(), q), 1), sd["conv3.weight"], sd["conv3.bias"]))), k
            hp = "proposal_generator.centernet_head."
            tw = B(f"tower{l + 3}", h, w)
            assert _ulp_close(tw, R.dense_conv(pos.float(), sd[hp + "bbox_tower.0.weight"], sd[hp + "bbox_tower.0.bias"], padding=1)), k
            t = F.relu(F.group_norm(tw.float(), 32, sd[hp + "bbox_tower.1.weight"], sd[hp + "bbox_tower.1.bias"], eps=1e-5)).bfloat16().float()
            hm = R.dense_conv(t, sd[hp + "agn_hm.weight"], sd[hp + "agn_hm.bias"], padding=1)
            rg = F.relu(R.dense_conv(t, sd[hp + "bbox_pred.weight"], sd[hp + "bbox_pred.bias"], padding=1) * sd[hp + f"scales.{l}.scale"])
        hd = e.buffer(f"head{l + 3}", (1, h, w)).cpu()
        # the normalised tower is itself a rounded tensor: a 1-ulp flip of one of its values moves the fp32 head output by ~1e-3 of its scale
        assert rel_err(hd[:, 4:5].numpy(), hm.numpy()) < 2e-3 and rel_err(hd[:, :4].numpy(), rg.numpy()) < 2e-3, k
    e.close()


@pytest.mark.gpu
def test_engine_bf16_storage_vs_oracle_and_fp32_tail(ore):
    """Whole eval hot path at the BASELINE shape in the bf16 STORAGE mode: feature maps against the oracle's restatement of the mode
    (independent bf16 rounding noise: the same bound as the operand mode), closer to it than to the fp32 network; the detection tail
    is fp32 and stays BIT-EXACT against ref_decode.c on the engine's own head outputs; eager and graph replays agree bit for bit."""
    sd = R.synth_state_dict(0)
    e = _bf16s_engine(ore, sd, (640, 640))
    img = R.synth_image(0)
    with R.operand_precision("bf16s"):
        ref = R.eval_dense(img, sd, R.synth_support(0))
    ref32 = R.eval_dense(img, sd, R.synth_support(0))
    first = None
    for use_graph in (False, True, True):
        e.eval_forward(img.cuda(), use_graph=use_graph)
        torch.cuda.synchronize()
        rms = lambda a, b: float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))      # noqa: E731
        for l, k in enumerate(("p3", "p4", "p5")):
            s = 640 >> (l + 3)
            got = e.buffer(k, (1, s, s)).float().cpu().numpy()
            assert rel_err(got, ref["features"][k].numpy()) < 3 * TOL_NET, k
            assert 1.5 * rms(got, ref["features"][k].numpy()) < rms(got, ref32["features"][k].numpy()), k
        hms, regs, heads = [], [], []
        for l in range(3):
            s = 640 >> (l + 3)
            hd = e.buffer(f"head{l + 3}").cpu().numpy().reshape(s, s, 5)
            heads.append(hd.copy())
            hms.append(np.ascontiguousarray(hd[..., 4]))
            regs.append(np.ascontiguousarray(hd[..., :4]))
        want = odec.decode_nms(hms, regs, (8, 16, 32), 1e-5, 1000, 0.6, 256)
        boxes, scores, keep = e.proposals()
        assert np.array_equal(keep.cpu().numpy(), want["keep"]) and len(want["keep"]) > 0
        assert np.array_equal(boxes.cpu().numpy(), want["boxes"]) and np.array_equal(scores.cpu().numpy(), want["scores"])
        if first is None:
            first = heads
        else:
            assert all(np.array_equal(a, b) for a, b in zip(first, heads))
    e.close()
