"""Size-independent properties at BASELINE.json's full sizes (no oracle needed): what must hold for ANY input.
NMS: idempotent, survivors pairwise below the threshold, every suppressed box overlaps an earlier survivor, order by score.
Top-k / decode: per-level counts, threshold, score monotonicity.  ROIAlign and conv: linearity.  Engine: determinism."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oh():
    import orehip
    orehip.lib()
    return orehip


def _iou(a, b):
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b - inter)


@pytest.mark.parametrize("n,thr", [(6000, 0.9), (2400, 0.6), (700, 0.3)])
def test_nms_properties_full_size(oh, n, thr):
    g = torch.Generator().manual_seed(n)
    ctr = torch.rand(n, 2, generator=g) * 640
    wh = torch.exp(torch.rand(n, 2, generator=g) * 2.5 + 2.0)
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).cuda()
    scores = torch.rand(n, generator=g).cuda()
    m7 = len(scores[1::7])
    scores[::7][:m7] = scores[1::7]                                        # plenty of exact score ties
    keep = oh.nms(boxes, scores, thr)
    kb, ks = boxes[keep], scores[keep]
    assert torch.all(ks[:-1] >= ks[1:])                                    # emitted in descending score order
    iou = _iou(kb, kb)
    iou.fill_diagonal_(0)
    assert float(iou.max()) <= thr                                         # survivors are pairwise compatible
    keep2 = oh.nms(kb, ks, thr)
    assert torch.equal(keep2.cpu(), torch.arange(len(keep)))               # idempotent
    dropped = torch.ones(n, dtype=torch.bool, device="cuda")
    dropped[keep] = False
    di = torch.nonzero(dropped).squeeze(1)
    if len(di):                                                            # every dropped box is covered by a survivor that outranks it
        cover = _iou(boxes[di], kb) > thr
        outranks = (ks[None, :] > scores[di][:, None]) | ((ks[None, :] == scores[di][:, None]) & (keep[None, :] < di[:, None]))
        assert bool((cover & outranks).any(1).all())


def test_detect_properties_full_size(oh):
    g = torch.Generator().manual_seed(3)
    heads = []
    for s in (80, 40, 20):
        h = torch.zeros(s, s, 8)
        h[..., :4] = torch.rand(s, s, 4, generator=g) * 6
        h[..., 4] = torch.randn(s, s, generator=g) * 2.5 - 1.0
        heads.append(h.cuda())
    for (pre, thr, post) in ((1000, 0.6, 256), (4000, 0.9, 2000)):
        o = oh.detect(heads, (8, 16, 32), 1e-5, pre, thr, post)
        n_pre, n_keep = int(o["counts"][0]), int(o["counts"][1])
        lv = o["pre_level"][:n_pre].cpu()
        for l, s in enumerate((80, 40, 20)):
            sig = torch.sigmoid(heads[l][..., 4]).flatten()
            want = min(int((sig > 1e-5).sum()), pre)
            assert int((lv == l).sum()) == want                              # exactly min(#candidates, pre_topk) per level
        sc = o["pre_scores"][:n_pre]
        assert float(sc.min()) > (1e-5) ** 0.5 * 0.999                       # score = sqrt(heatmap) of a candidate above the threshold
        out_s = o["out_scores"][:n_keep]
        assert torch.all(out_s[:-1] >= out_s[1:])
        if n_keep > post:                                                   # more than post_topk only through ties at the k-th score
            assert float(out_s[post - 1]) == float(out_s[-1])
        b = o["out_boxes"][:n_keep]
        assert bool((b[:, 2] > b[:, 0]).all() and (b[:, 3] > b[:, 1]).all())


def test_roi_align_and_conv_linearity_full_size(oh):
    g = torch.Generator().manual_seed(8)
    f1 = [torch.randn(1, 80 >> l, 80 >> l, 128, generator=g).cuda() for l in range(3)]
    f2 = [torch.randn(1, 80 >> l, 80 >> l, 128, generator=g).cuda() for l in range(3)]
    ctr = torch.rand(256, 2, generator=g) * 640
    wh = torch.exp(torch.rand(256, 2, generator=g) * 3 + 2.5)
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).cuda()
    a, b = 0.75, -1.5
    r1, r2 = oh.roi_align(f1, boxes, (8, 16, 32)), oh.roi_align(f2, boxes, (8, 16, 32))
    r12 = oh.roi_align([a * x + b * y for x, y in zip(f1, f2)], boxes, (8, 16, 32))
    assert float((r12 - (a * r1 + b * r2)).abs().max()) <= 2e-5 * float(r12.abs().max())
    # stem_2-sized 3x3 conv (weight-stationary kernel) and a stage-5 sized one (K-split tiles): linear in the input, no epilogue
    for (H, W, Cin, Cout) in ((320, 320, 64, 64), (20, 20, 384, 112)):
        x1, x2 = torch.randn(1, H, W, Cin, generator=g).cuda(), torch.randn(1, H, W, Cin, generator=g).cuda()
        w = oh.pack_conv_weight(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda()
        y1, y2 = oh.conv2d(x1, w, Cout, 3), oh.conv2d(x2, w, Cout, 3)
        y12 = oh.conv2d(a * x1 + b * x2, w, Cout, 3)
        assert float((y12 - (a * y1 + b * y2)).abs().max()) <= 3e-5 * float(y12.abs().max())
        assert torch.equal(oh.conv2d(x1, w, Cout, 3), y1)                   # and bit-reproducible


def test_winograd_kernels_linearity_full_size(oh):
    """The Winograd kernels at the path's full sizes without an oracle: linear in the input, bit-reproducible, and equal to the direct
    kernels within the fp32 rounding of the transforms -- stem_2 (64 ch), stage-2 layer 0 (128 ch), stage 3 (80 / 112 ch: the nu-split
    build), and the three FPN output convs as one per-level launch against three single-level launches."""
    g = torch.Generator().manual_seed(18)
    a, b = 0.75, -1.5
    for (H, W, Cin, Cout) in ((320, 320, 64, 64), (160, 160, 128, 64), (80, 80, 80, 80), (80, 80, 112, 80)):
        x1, x2 = torch.randn(1, H, W, Cin, generator=g).cuda(), torch.randn(1, H, W, Cin, generator=g).cuda()
        w = oh.pack_conv_weight(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda()
        U = oh.winograd_weight(w, Cout, Cin)
        y1, y2 = oh.conv2d(x1, w, Cout, 3, w_wino=U), oh.conv2d(x2, w, Cout, 3, w_wino=U)
        y12 = oh.conv2d(a * x1 + b * x2, w, Cout, 3, w_wino=U)
        d1 = oh.conv2d(x1, w, Cout, 3)
        assert not torch.equal(y1, d1), "the Winograd kernel did not run"
        assert float((y12 - (a * y1 + b * y2)).abs().max()) <= 3e-5 * float(y12.abs().max())
        assert float((y1 - d1).abs().max()) <= 2e-5 * float(d1.abs().max())
        assert torch.equal(oh.conv2d(x1, w, Cout, 3, w_wino=U), y1)
    HW = [(80, 80), (40, 40), (20, 20)]
    rows = torch.randn(sum(h * w_ for h, w_ in HW), 128, generator=g).cuda()
    ws = [oh.pack_conv_weight(torch.randn(128, 128, 3, 3, generator=g) * 0.03).cuda() for _ in HW]
    Us = torch.stack([oh.winograd_weight(w, 128, 128) for w in ws]).contiguous()
    bias = (torch.randn(3, 128, generator=g) * 0.1).cuda()
    y = oh.conv2d_levels(rows, HW, 1, ws[0], 128, 3, shift=bias, ep_stride=128, w_wino=Us, w_wino_level_stride=Us.shape[1])
    r0 = 0
    oh.lib().ore_conv_set_plan_override(-7, 2, 0, 0, 0)            # single-level reference launches on the Winograd kernel whatever the row count
    try:
        for l, (h, w_) in enumerate(HW):
            yl = oh.conv2d(rows[r0:r0 + h * w_].reshape(1, h, w_, 128), ws[l], 128, 3, shift=bias[l].contiguous(), w_wino=Us[l].contiguous())
            assert torch.equal(y[r0:r0 + h * w_], yl.reshape(-1, 128)), "level %d of the grouped launch differs from its own launch" % l
            r0 += h * w_
    finally:
        oh.lib().ore_conv_set_plan_override(-7, 1, 0, 0, 0)


def test_descriptor_kernels_properties_full_size(oh):
    """k_conv_gd / k_conv_kd at the 640x640 shapes the plan gives them, without an oracle: the automatic plan does not run the round-3
    kernel (switching both families off changes bits somewhere), the result is linear in the input, bit-reproducible, within fp32
    rounding of the round-3 kernel, and the eSE column sums a launch emits add up to the column sums of its output."""
    g = torch.Generator().manual_seed(28)
    a, b = 0.75, -1.5
    L = oh.lib()
    shapes = [(320, 320, 64, 128, 3, 2),      # stem_3                      -> k_conv_gd 112 x 128, eight waves
              (160, 160, 320, 112, 1, 1),     # stage-2 concat              -> k_conv_gd 128 x 64, eight waves
              (80, 80, 352, 256, 1, 1),       # stage-3 concat              -> k_conv_gd 64 x 64
              (80, 80, 256, 128, 1, 1),       # lateral 3 (without its add) -> k_conv_gd 64 x 64, eight waves
              (40, 40, 256, 96, 3, 1),        # stage 4, layer 0            -> k_conv_kd
              (20, 20, 112, 112, 3, 1),       # stage 5, layers 1 / 2       -> k_conv_kd
              (20, 20, 720, 512, 1, 1)]       # stage-5 concat              -> k_conv_kd
    differs = 0
    for (H, W, Cin, Cout, k, st) in shapes:
        x1, x2 = torch.randn(1, H, W, Cin, generator=g).cuda(), torch.randn(1, H, W, Cin, generator=g).cuda()
        w = oh.pack_conv_weight(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).cuda()
        y1, y2 = oh.conv2d(x1, w, Cout, k, st), oh.conv2d(x2, w, Cout, k, st)
        y12 = oh.conv2d(a * x1 + b * x2, w, Cout, k, st)
        assert float((y12 - (a * y1 + b * y2)).abs().max()) <= 3e-5 * float(y12.abs().max())
        assert torch.equal(oh.conv2d(x1, w, Cout, k, st), y1)
        yc, cs = oh.conv2d(x1, w, Cout, k, st, want_colsum=True)              # (the plan may pick another kernel when column sums are asked for)
        ref_cs = yc.double().sum((0, 1, 2))
        assert float((cs.double().sum(0)[:Cout] - ref_cs).abs().max()) <= 2e-5 * float(ref_cs.abs().max() + yc.abs().sum() / Cout * 1e-2)
        assert float((yc - y1).abs().max()) <= 2e-5 * float(y1.abs().max())
        L.ore_conv_set_plan_override(-14, 0, 0, 0, 0); L.ore_conv_set_plan_override(-12, 0, 0, 0, 0)
        try:
            y_old = oh.conv2d(x1, w, Cout, k, st)
        finally:
            L.ore_conv_set_plan_override(-14, 1, 0, 0, 0); L.ore_conv_set_plan_override(-12, 1, 0, 0, 0)
        assert float((y1 - y_old).abs().max()) <= 2e-5 * float(y_old.abs().max())
        differs += int(not torch.equal(y1, y_old))
    assert differs >= 3, "the descriptor kernels did not run (the plan fell back to the round-3 kernels everywhere)"


def test_centernet_targets_properties_full_size(oh):
    """640x640, 128 boxes: every positive index addresses the cell containing its box centre on a level that cares for the box size;
    a location with a regression target lies inside that box; heat-map in [0, 1] and exactly 1 at positive cells."""
    g = torch.Generator().manual_seed(6)
    ctr = torch.rand(128, 2, generator=g) * 600 + 20
    wh = torch.exp(torch.rand(128, 2, generator=g) * 3.0 + 2.3)
    gt = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).clamp(0, 640)
    shapes = [(80, 80), (40, 40), (20, 20)]
    o = oh.centernet_targets([gt], shapes)
    n = int(o["pos_count"])
    pos = o["pos_inds"][:n].cpu()
    hm, reg = o["hm_targets"].cpu(), o["reg_targets"].cpu()
    assert float(hm.min()) >= 0 and float(hm.max()) <= 1.0 and n >= 128
    assert torch.all(hm[pos] == 1.0)                                        # the discretised centre is a peak of the Gaussian
    has = reg.max(1)[0] >= 0
    assert bool((reg[has] > 0).all()) and int(has.sum()) > 0                # targets only strictly inside a box
    row0 = [0, 6400, 8000]
    cx, cy = (gt[:, 0] + gt[:, 2]) / 2, (gt[:, 1] + gt[:, 3]) / 2
    cells = set()
    for l, s in enumerate((8, 16, 32)):
        W = 640 // s
        for i in range(128):
            cells.add(row0[l] + int(cy[i] / s) * W + int(cx[i] / s))
    assert set(pos.tolist()) <= cells


@pytest.mark.parametrize("n", [1, 63, 64, 65, 127, 129, 1000, 1537, 2047, 2048, 2049, 3000, 3071, 3072])
def test_column_scan_vs_oracle_every_ring_position(oh, n):
    """k_nms_mask_t + k_nms_scan_t (round 5: one consumer wave, 15 producer waves, a packed LDS ring of column blocks) against the oracle's
    NMS (oracle/ref_decode.c), keep lists bit for bit, at row counts that end inside / at / behind a 64-row block and that fill all 48
    column blocks: the ring wraps after column 19 and every later column waits for the consumer's progress (the `need` table) -- the
    bench shape stops at 38 columns.  Boxes are clustered (a few hundred centres, jittered), so most rows are suppressed by an earlier
    survivor, the in-block fixpoint has chains to walk and score ties are plentiful."""
    import numpy as np
    from oracle import decode as odec
    g = torch.Generator().manual_seed(1000 + n)
    centres = torch.rand(max(n // 12, 1), 2, generator=g) * 600 + 20
    idx = torch.randint(0, len(centres), (n,), generator=g)
    ctr = centres[idx] + torch.randn(n, 2, generator=g) * 3.0
    wh = torch.exp(torch.rand(n, 2, generator=g) * 0.6 + 3.2)
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).contiguous()
    scores = (torch.rand(n, generator=g) * 0.98 + 0.01).contiguous()
    m5 = len(scores[1::5])
    scores[::5][:m5] = scores[1::5]                                         # exact ties
    for thr in (0.6, 0.9):
        want = odec.nms(boxes.numpy(), scores.numpy(), thr)
        got = oh.nms(boxes.cuda(), scores.cuda(), thr).cpu().numpy()
        assert np.array_equal(got, want), (n, thr, len(got), len(want))
        assert 0 < len(want) <= n


def test_multi_weight_repack_equals_one_by_one(oh):
    """ore_pack_conv_weights_multi_fwd (one launch for all the trainable weights of a step) writes, bit for bit, what
    ore_pack_conv_weight_fwd writes weight by weight -- forward and data-gradient layouts, 1x1 and 3x3, widths that are not multiples
    of 16 on the data-gradient side -- and the forward layout is the host packer's (ore_pack_conv_weight_host)."""
    g = torch.Generator().manual_seed(11)
    shapes = [(96, 256, 3, 3), (384, 544, 1, 1), (112, 112, 3, 3), (128, 512, 1, 1), (4, 128, 3, 3), (1, 128, 3, 3), (128, 128, 3, 3)]
    jobs, want = [], []
    for (co, ci, kh, kw) in shapes:
        w = torch.randn(co, ci, kh, kw, generator=g).cuda()
        for dgrad in (False, True):
            ref = oh.pack_conv_weight_dev(w, dgrad=dgrad)
            out = torch.full_like(ref, float("nan"))
            jobs.append((w, dgrad, out))
            want.append(ref)
        assert torch.equal(want[-2].cpu(), oh.pack_conv_weight(w.cpu()))
    oh.pack_conv_weights_multi(jobs)
    torch.cuda.synchronize()
    for (w, dgrad, out), ref in zip(jobs, want):
        assert torch.equal(out, ref), (tuple(w.shape), dgrad)
    # the cached job table follows the CONTENTS of the masters: new values through the same pointers are repacked by the same table
    for w, _, _ in jobs[::2]:                                               # every master appears twice (forward + data-gradient job)
        w.mul_(-0.5)
    oh.pack_conv_weights_multi(jobs)
    for (w, dgrad, out), ref in zip(jobs, want):
        assert torch.equal(out, ref * -0.5)


def test_flop_counter_counts_algorithmic_flops_of_the_entry_points(oh):
    """ore_flop_counter_read: 2*M*Cout*Cin*kh*kw per forward conv call, whatever kernel the plan picks (Winograd executes fewer
    multiplies, split-K more launches: neither changes the count) -- the figure bench.py's training roofline is priced with."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 40, 40, 96, generator=g).cuda()
    w = torch.randn(96, 96, 3, 3, generator=g)
    wp = oh.pack_conv_weight(w).cuda()
    oh.flop_counter(reset=True)
    assert oh.flop_counter() == (0.0, 0)
    oh.conv2d(x, wp, 96, 3)
    f1, n1 = oh.flop_counter()
    assert n1 == 1 and f1 == 2.0 * (2 * 40 * 40) * 96 * 96 * 9
    oh.conv2d(x, wp, 96, 3, w_wino=oh.winograd_weight(wp, 96, 96) if oh.winograd_covers(96, 96) else None)
    oh.conv2d(x, wp, 96, 3, stride=2)
    f3, n3 = oh.flop_counter(reset=True)
    assert n3 == 3 and f3 == 2 * f1 + 2.0 * (2 * 20 * 20) * 96 * 96 * 9
    assert oh.flop_counter() == (0.0, 0)
