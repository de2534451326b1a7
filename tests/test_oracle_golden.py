"""Pin the CPU oracle (oracle/ref_model.py, oracle/ref_decode.c) against golden vectors produced by
executing the reference's own files (oracle/refrun/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import decode as odec
from oracle import ref_model as R

T = torch.from_numpy


@pytest.fixture(scope="module")
def sd():
    return R.synth_state_dict(0)


def test_backbone_fpn(golden, sd):
    g = golden("backbone_fpn_96x128")
    out = R.backbone_fpn(T(g["x"]), sd)
    for k in ("p3", "p4", "p5"):
        assert out[k].shape == g[k].shape
        assert rel_err(out[k].numpy(), g[k]) < 1e-5, k


def test_conv_blocks_and_stages(golden, sd):
    p = "backbone.bottom_up."
    g = golden("conv_stem2")
    assert rel_err(R.conv_bn_relu(T(g["x"]), sd, p + "stem.stem_2", 1, 1).numpy(), g["y"]) < 1e-5
    g = golden("conv_stem3_odd")
    assert rel_err(R.conv_bn_relu(T(g["x"]), sd, p + "stem.stem_3", 2, 1).numpy(), g["y"]) < 1e-5
    for name, k in (("osa_stage3_odd", 3), ("osa_stage5", 5)):
        g = golden(name)
        x = torch.nn.functional.max_pool2d(T(g["x"]), 3, 2, ceil_mode=True)
        y = R.osa_module(x, sd, f"{p}stage{k}.OSA{k}_1.", f"OSA{k}_1", 3, False)
        assert y.shape == g["y"].shape
        assert rel_err(y.numpy(), g["y"]) < 1e-5, name


def test_preprocess(golden):
    g = golden("preprocess_75x100")
    x = R.preprocess(T(g["image"]))
    assert x.shape == g["x"].shape
    assert np.array_equal(x.numpy(), g["x"])


def test_sm_block(golden, sd):
    for lvl, seg in ((3, 32), (5, 8)):
        g = golden(f"sm_block_p{lvl}")
        y = R.sm_block(T(g["x"]), sd, f"vip_p{lvl}.", seg)
        assert rel_err(y[:1].numpy(), g["y0"]) < 1e-5
        assert rel_err(y.permute(0, 3, 2, 1).mean(0, True).numpy(), g["proto"]) < 1e-5


def test_correlation(golden, sd):
    g = golden("correlation")
    for k in ("p3", "p4", "p5"):
        out = R.correlation(T(g["q_" + k]), T(g["s_" + k]), sd["conv3.weight"], sd["conv3.bias"])
        assert rel_err(out.numpy(), g["out_" + k]) < 1e-6, k


def test_centernet_head(golden, sd):
    g = golden("cn_head")
    regs, hms = R.centernet_head([T(g[f"x{l}"]) for l in range(3)], sd)
    for l in range(3):
        assert rel_err(regs[l].numpy(), g[f"reg{l}"]) < 1e-5
        assert rel_err(hms[l].numpy(), g[f"hm{l}"]) < 1e-5


def test_sigmoid_vs_torch():
    x = np.linspace(-40, 40, 400001).astype(np.float32)
    y = odec.sigmoid(x)
    t = torch.sigmoid(T(x)).numpy()
    ulp = np.abs(y.view(np.int32).astype(np.int64) - t.view(np.int32).astype(np.int64))
    assert ulp[t > 1e-30].max() <= 4
    assert (np.diff(y) >= 0).all()  # monotone: ordering of logits is preserved


@pytest.mark.parametrize("tag", ["sparse", "dense"])
def test_decode_nms_vs_reference_run(golden, tag):
    """The reference's CenterNet.inference (executed) vs ref_decode.c on the same head outputs.

    The reference sigmoid is torch's (<=2 ulp from ours), so scores/boxes are compared with a 1e-6
    tolerance and index sets by matching boxes; the selected SETS and NMS keep lists must agree."""
    g = golden(f"cn_infer_640_{tag}")
    r = odec.decode_nms([g[f"hm{l}"] for l in range(3)], [g[f"reg{l}"] for l in range(3)], (8, 16, 32),
                        1e-5, 1000, 0.6, 256)
    # pre-NMS candidates: same count; reference order within a level is topk(sorted=False) order ->
    # compare as sets of (box) rows
    assert len(r["pre_scores"]) == len(g["pre_scores"])

    def rows(b, s):
        a = np.concatenate([b, s[:, None]], 1).astype(np.float64)
        return a[np.lexsort(a.T[::-1])]

    np.testing.assert_allclose(rows(r["pre_boxes"], r["pre_scores"]), rows(g["pre_boxes"], g["pre_scores"]),
                               rtol=2e-6, atol=1e-6)
    # final proposals: same order (descending score), same boxes
    assert r["boxes"].shape == g["boxes"].shape
    np.testing.assert_allclose(r["scores"], g["scores"], rtol=2e-6)
    np.testing.assert_allclose(r["boxes"], g["boxes"], rtol=2e-6, atol=1e-5)


def test_nms_c_vs_numpy_edge_cases():
    rng = np.random.default_rng(0)
    cases = []
    # ties in score, IoU exactly at threshold, zero-area, 0/1 boxes
    cases.append((np.zeros((0, 4), np.float32), np.zeros(0, np.float32)))
    cases.append((np.array([[0, 0, 10, 10]], np.float32), np.array([0.5], np.float32)))
    cases.append((np.array([[0, 0, 10, 10], [0, 0, 10, 10], [0, 0, 10, 5], [20, 20, 20, 20]], np.float32),
                  np.array([0.5, 0.5, 0.5, 0.9], np.float32)))
    b = rng.uniform(0, 100, (300, 2)).astype(np.float32)
    wh = rng.uniform(5, 40, (300, 2)).astype(np.float32)
    cases.append((np.concatenate([b, b + wh], 1), np.round(rng.uniform(0, 1, 300), 2).astype(np.float32)))
    for boxes, scores in cases:
        for thr in (0.5, 0.6, 0.9):
            assert np.array_equal(odec.nms(boxes, scores, thr), odec.nms_numpy(boxes, scores, thr))
    # IoU exactly 0.5 is NOT suppressed at thr 0.5 (strict >)
    boxes = np.array([[0, 0, 2, 1], [1, 0, 3, 1]], np.float32)  # inter 1, union 3 -> 1/3
    boxes2 = np.array([[0, 0, 2, 2], [0, 0, 2, 1]], np.float32)  # inter 2, union 4 -> 0.5
    assert list(odec.nms(boxes2, np.array([0.9, 0.8], np.float32), 0.5)) == [0, 1]
    assert list(odec.nms(boxes2, np.array([0.9, 0.8], np.float32), 0.49)) == [0]
    assert list(odec.nms(boxes, np.array([0.1, 0.8], np.float32), 0.3)) == [1]


def test_centernet_train_targets_and_losses(golden):
    """a12: the restated ground truth / losses equal the executed reference (fsod_rpn.py:702-779, 803-1065) on B=2."""
    g = golden("cn_train_targets")
    shapes = [tuple(int(v) for v in s) for s in g["shapes"]]
    pos, reg, hm = R.centernet_targets([torch.from_numpy(g["gt0"]), torch.from_numpy(g["gt1"])], shapes)
    assert np.array_equal(pos.numpy(), g["pos_inds"])
    assert np.array_equal(reg.numpy(), g["reg_targets"])
    assert np.array_equal(hm.numpy(), g["hms"])
    ls = R.centernet_losses(torch.from_numpy(g["reg_pred"]), torch.from_numpy(g["hm_logit"]), pos, reg, hm)
    for k, ref in (("loss_centernet_loc", "loss_loc"), ("loss_centernet_agn_pos", "loss_pos"), ("loss_centernet_agn_neg", "loss_neg")):
        assert abs(float(ls[k]) - float(g[ref])) <= 1e-6 * abs(float(g[ref]))


def test_roi_train_pieces_match_reference_run(golden):
    """Second-stage training helpers of the oracle (oracle/ref_train.py) against the vendored detectron2's own pairwise_iou,
    Matcher([0.6],[0,1]), subsample_labels and Box2BoxTransform.get_deltas, executed by oracle/refrun/gen_golden.py."""
    import torch
    from oracle import ref_train as T
    g = golden("roi_train_pieces")
    gt, boxes = torch.from_numpy(g["gt"]), torch.from_numpy(g["boxes"])
    n_g = gt.shape[0]
    iou = T.pairwise_iou(gt, boxes)
    np.testing.assert_array_equal(iou.numpy(), g["iou"])
    allb, matched, labels = T.label_proposals(boxes[:-n_g], gt, 0.6)
    np.testing.assert_array_equal(allb.numpy(), g["boxes"])
    np.testing.assert_array_equal(matched.numpy(), g["matched_idx"])
    np.testing.assert_array_equal(labels.numpy(), g["labels"])
    torch.manual_seed(int(g["seed"]))
    sampled = T.sample_labels(labels, 128, 0.5, lambda n: torch.randperm(n))
    np.testing.assert_array_equal(sampled.numpy(), g["sampled"])
    fg = sampled[labels[sampled] == 0]
    np.testing.assert_array_equal(fg.numpy(), g["fg_rows"])
    d = T.get_deltas(allb[fg], gt[matched[fg]])
    np.testing.assert_allclose(d.numpy(), g["deltas"], rtol=1e-6, atol=1e-6)
