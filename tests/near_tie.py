"""Explaining the rows on which two correct implementations of "sort by score, greedy NMS, keep the best k" may differ.

The reference's greedy NMS (ref:CenterNet2/centernet/modeling/layers/ml_nms.py:4-31 -> d2z:layers/nms.py:10-30 -> torchvision `nms`:
suppress when IoU > thr, IoU = inter / (a_i + a_j - inter)) and its post-NMS filter (ref:fewx/modeling/fsod/fsod_rpn.py:1198-1206:
score >= k-th score) are chains of hard decisions on fp32 inputs.  Two fp32 pipelines whose feature maps agree to 1e-6 hand the
NMS boxes that agree to ~1e-4 px, so a pair whose IoU sits within ~1e-5 of the threshold, or a row whose score sits on the k-th
place, can legitimately be decided differently -- and one flipped decision changes which later rows are suppressed (a cascade that is
itself not a near-tie).  `guided_nms_explain` therefore does not compare outputs row by row: it re-runs the greedy walk on OUR
candidate list and lets the REFERENCE's output decide only the decisions that are near-ties (|IoU - thr| <= iou_band against some kept
row and no kept row clearly above thr; |score - cut| <= score_band * cut (5e-6) for the top-k cut).  Every clear decision is taken as computed.
If the walk then reproduces the reference's output exactly, every difference between the two outputs is explained by a near-tie
(directly or as its cascade); anything else is returned in `unexplained`.

Test infrastructure (used by tests/ only)."""
import numpy as np


def iou_one_to_many(b, others):
    """torchvision's IoU of one box against many, evaluated in float64 on the fp32 values."""
    b = np.asarray(b, np.float64)
    o = np.asarray(others, np.float64).reshape(-1, 4)
    iw = np.clip(np.minimum(b[2], o[:, 2]) - np.maximum(b[0], o[:, 0]), 0, None)
    ih = np.clip(np.minimum(b[3], o[:, 3]) - np.maximum(b[1], o[:, 1]), 0, None)
    inter = iw * ih
    union = (b[2] - b[0]) * (b[3] - b[1]) + (o[:, 2] - o[:, 0]) * (o[:, 3] - o[:, 1]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        v = inter / union
    return np.where(union > 0, v, 0.0)


def match_rows(cand_boxes, cand_scores, ref_boxes, ref_scores, box_tol, score_rtol):
    """For every reference row the candidate it is (nearest in max-abs box distance, inside box_tol and score_rtol), or -1.
    A candidate is claimed by at most one reference row (the closest)."""
    cand_boxes, ref_boxes = np.asarray(cand_boxes, np.float64).reshape(-1, 4), np.asarray(ref_boxes, np.float64).reshape(-1, 4)
    out = np.full(len(ref_boxes), -1, dtype=np.int64)
    if len(cand_boxes) == 0:
        return out
    taken = {}
    for r in range(len(ref_boxes)):
        d = np.abs(cand_boxes - ref_boxes[r]).max(1)
        ds = np.abs(np.asarray(cand_scores, np.float64) - float(ref_scores[r]))
        d = np.where(ds <= score_rtol * abs(float(ref_scores[r])) + 1e-12, d, np.inf)
        j = int(d.argmin())
        if d[j] <= box_tol and (j not in taken or d[j] < taken[j][1]):
            if j in taken:
                out[taken[j][0]] = -1
            taken[j] = (r, d[j])
            out[r] = j
    return out


def guided_nms_explain(cand_boxes, cand_scores, ref_boxes, ref_scores, nms_thr, post_topk=None, *, iou_band=2e-5, score_band=5e-6,
                       box_tol=2e-2, score_rtol=2e-4, max_out=None, drop_empty=False):
    """cand_*: OUR rows entering the NMS (any order; the walk sorts them by score, descending, stable).  ref_*: the REFERENCE's output.
    post_topk: the score >= k-th filter behind the NMS (None: none).  max_out: a plain truncation of the kept list (top-k by rank,
    the second stage's `[:topk]`).  drop_empty: the walk's output then loses its empty boxes (detector_postprocess,
    d2z:modeling/postprocessing.py:10-75, after the truncation) before it is compared.  Returns {"unexplained": [...], "ambiguous": n, "kept": indices into cand}.  Empty `unexplained` <=>
    the reference's output is reachable from our candidates by deciding only near-ties its way."""
    cb = np.asarray(cand_boxes, np.float32).reshape(-1, 4)
    cs = np.asarray(cand_scores, np.float32).reshape(-1)
    which = match_rows(cb, cs, ref_boxes, ref_scores, box_tol, score_rtol)
    unexplained = [("reference row has no candidate here", int(r)) for r in np.where(which < 0)[0]]
    in_ref = np.zeros(len(cb), dtype=bool)
    in_ref[which[which >= 0]] = True
    order = np.argsort(-cs.astype(np.float64), kind="stable")
    kept, ambiguous = [], 0
    for c in order:
        if kept:
            iou = iou_one_to_many(cb[c], cb[kept])
            hi, lo = bool((iou > nms_thr + iou_band).any()), bool((iou <= nms_thr - iou_band).all())
        else:
            hi, lo = False, True
        if hi:
            if in_ref[c]:
                unexplained.append(("kept by the reference, clearly suppressed here", int(c)))
            continue
        if lo:
            kept.append(int(c))
            continue
        ambiguous += 1                                   # only near-tie overlaps stand between this row and the output: the reference decides
        if in_ref[c]:
            kept.append(int(c))
    # behind the NMS: the k-th-score cut (ties at the cut may keep more than k) or a plain truncation
    final = list(kept)
    if post_topk is not None and len(kept) > post_topk:
        sc = np.sort(cs[kept].astype(np.float64))[::-1]
        cut = sc[post_topk - 1]
        final = []
        for c in kept:
            s = float(cs[c])
            if abs(s - cut) <= score_band * abs(cut):
                ambiguous += 1
                if in_ref[c]:
                    final.append(c)
            elif s > cut:
                final.append(c)
    if max_out is not None and len(final) > max_out:
        # rank truncation: rows whose score ties the last admitted one (within score_band) may swap across the cut
        edge = float(cs[final[max_out - 1]])
        head = [c for c in final[:max_out]]
        tail = [c for c in final[max_out:] if abs(float(cs[c]) - edge) <= score_band * abs(edge) and in_ref[c]]
        drop = [c for c in head if abs(float(cs[c]) - edge) <= score_band * abs(edge) and not in_ref[c]]
        final = [c for c in head if c not in drop] + tail
        ambiguous += len(tail) + len(drop)
    if drop_empty:
        final = [c for c in final if cb[c, 2] > cb[c, 0] and cb[c, 3] > cb[c, 1]]
    fin = set(final)
    for c in final:
        if not in_ref[c]:
            unexplained.append(("kept here on clear decisions, absent from the reference", int(c)))
    for c in np.where(in_ref)[0]:
        if int(c) not in fin and not any(u[1] == int(c) and u[0].startswith("kept by the reference") for u in unexplained):
            unexplained.append(("in the reference, dropped here by the cut on a clear margin", int(c)))
    return {"unexplained": unexplained, "ambiguous": ambiguous, "kept": final}
