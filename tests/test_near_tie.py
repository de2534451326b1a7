"""tests/near_tie.py (the explanation of "as a set" differences between two greedy-NMS outputs) on constructed cases, with the oracle's
NMS (oracle/ref_decode.c, the restated torchvision algorithm) producing both sides.  CPU only."""
import numpy as np

from near_tie import guided_nms_explain
from oracle import decode as odec


def _boxes_with_a_tie(eps):
    """A (0.9) and B (0.8) overlap with IoU = 0.6 -+ eps; C (0.7) is clearly suppressed by B (and untouched by A); D is far away.
    eps > 0: B survives and removes C; eps < 0: A removes B and C survives -- one near-tie decision, a two-row cascade."""
    # A = [0,0,100,100]; B = [0,y,100,100+y]: IoU = (100-y)/(100+y) = 0.6 -> y = 25
    y = 25.0 + eps
    a = [0.0, 0.0, 100.0, 100.0]
    b = [0.0, y, 100.0, 100.0 + y]
    c = [0.0, 45.0, 100.0, 140.0]           # IoU(B, C) = 0.70 > 0.6, IoU(A, C) = 0.39
    d = [300.0, 300.0, 340.0, 350.0]
    return np.array([a, b, c, d], np.float32), np.array([0.9, 0.8, 0.7, 0.6], np.float32)


def test_a_near_tie_and_its_cascade_are_explained():
    b_lo, s = _boxes_with_a_tie(+2e-4)      # IoU(A,B) just below 0.6: B kept, C suppressed
    b_hi, _ = _boxes_with_a_tie(-2e-4)      # just above: B suppressed, C kept
    k_lo, k_hi = odec.nms(b_lo, s, 0.6), odec.nms(b_hi, s, 0.6)
    assert list(k_lo) == [0, 1, 3] and list(k_hi) == [0, 2, 3]
    # "ours" = the lo side's candidates, "reference" = the hi side's output: rows B and C differ, one near-tie explains both
    r = guided_nms_explain(b_lo, s, b_hi[k_hi], s[k_hi], 0.6, iou_band=1e-5, box_tol=1e-2)
    assert r["unexplained"] == [] and r["ambiguous"] == 1 and sorted(r["kept"]) == [0, 2, 3]
    r = guided_nms_explain(b_hi, s, b_lo[k_lo], s[k_lo], 0.6, iou_band=1e-5, box_tol=1e-2)
    assert r["unexplained"] == [] and r["ambiguous"] == 1 and sorted(r["kept"]) == [0, 1, 3]
    # identical outputs: nothing ambiguous is needed
    r = guided_nms_explain(b_lo, s, b_lo[k_lo], s[k_lo], 0.6, iou_band=1e-7)
    assert r["unexplained"] == [] and sorted(r["kept"]) == [0, 1, 3]


def test_a_difference_that_is_not_a_tie_is_reported():
    b, s = _boxes_with_a_tie(+2.0)           # IoU(A,B) = 0.587: clearly below the threshold
    keep = odec.nms(b, s, 0.6)
    assert list(keep) == [0, 1, 3]
    wrong = np.array([0, 2, 3])              # a "reference" that dropped B and kept C without any tie to point at
    r = guided_nms_explain(b, s, b[wrong], s[wrong], 0.6, iou_band=1e-5)
    kinds = sorted(u[0] for u in r["unexplained"])
    assert kinds == ["kept by the reference, clearly suppressed here", "kept here on clear decisions, absent from the reference"], r
    moved = b[keep].copy()
    moved[1] += 0.5                          # a reference row that is none of our candidates (0.5 px off)
    r = guided_nms_explain(b, s, moved, s[keep], 0.6, box_tol=2e-2)
    assert ("reference row has no candidate here", 1) in r["unexplained"]


def test_the_kth_score_cut_and_rank_truncation():
    rng = np.random.default_rng(0)
    n = 40
    ctr = rng.uniform(0, 2000, (n, 2))
    b = np.concatenate([ctr, ctr + 20.0], 1).astype(np.float32)          # disjoint boxes: NMS keeps all
    s = np.sort(rng.uniform(0.1, 0.9, n).astype(np.float32))[::-1].copy()
    s[10] = s[9]                                                             # a tie on the 10th place: score >= kth keeps 11
    ref = np.arange(11)
    r = guided_nms_explain(b, s, b[ref], s[ref], 0.6, post_topk=10)
    assert r["unexplained"] == [] and len(r["kept"]) == 11
    ref = np.arange(10)                                                      # a reference that broke the tie the other way
    r = guided_nms_explain(b, s, b[ref], s[ref], 0.6, post_topk=10)
    assert r["unexplained"] == [] and r["ambiguous"] >= 1
    ref = np.array(list(range(9)) + [12])                                    # row 12 is clearly below the cut
    r = guided_nms_explain(b, s, b[ref], s[ref], 0.6, post_topk=10)
    assert any(u[0].startswith("kept here on clear decisions") or u[0].startswith("in the reference, dropped") for u in r["unexplained"])
    # plain rank truncation (the second stage's [:topk])
    s2 = np.sort(rng.uniform(0.1, 0.9, n).astype(np.float32))[::-1].copy()
    r = guided_nms_explain(b, s2, b[:5], s2[:5], 0.6, max_out=5)
    assert r["unexplained"] == [] and r["kept"] == [0, 1, 2, 3, 4]
    r = guided_nms_explain(b, s2, b[[0, 1, 2, 3, 7]], s2[[0, 1, 2, 3, 7]], 0.6, max_out=5)
    assert r["unexplained"] != []
