"""ctypes loader for oracle/ref_decode.c + a pure-numpy twin for small cases.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, List, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    so = os.path.join(_HERE, "_build", "liboracle_decode.so")
    src = os.path.join(_HERE, "ref_decode.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.oracle_sigmoid.restype = ctypes.c_float
        L.oracle_sigmoid.argtypes = [ctypes.c_float]
        L.oracle_nms.restype = ctypes.c_int64
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def sigmoid(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().oracle_sigmoid_array(_p(x, ctypes.c_float), _p(y, ctypes.c_float), ctypes.c_int64(x.size))
    return y


def nms(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    keep = np.empty(max(len(scores), 1), dtype=np.int64)
    n = lib().oracle_nms(_p(boxes, ctypes.c_float), _p(scores, ctypes.c_float), ctypes.c_int64(len(scores)),
                         ctypes.c_float(thr), _p(keep, ctypes.c_int64))
    assert n >= 0
    return keep[:n].copy()


def decode_nms(hm: Sequence[np.ndarray], reg: Sequence[np.ndarray], strides: Sequence[int],
               score_thresh: float = 1e-5, pre_topk: int = 1000, nms_thresh: float = 0.6,
               post_topk: int = 256) -> Dict[str, np.ndarray]:
    """hm[l]: [H,W] logits; reg[l]: [H,W,4] (after Scale+ReLU, stride units)."""
    L = len(hm)
    hm = [np.ascontiguousarray(h, dtype=np.float32) for h in hm]
    reg = [np.ascontiguousarray(r, dtype=np.float32) for r in reg]
    for h, r in zip(hm, reg):
        assert h.ndim == 2 and r.shape == h.shape + (4,), (h.shape, r.shape)
    H = np.array([h.shape[0] for h in hm], dtype=np.int32)
    W = np.array([h.shape[1] for h in hm], dtype=np.int32)
    S = np.array(list(strides), dtype=np.int32)
    cap = max(L * pre_topk, 1)
    pb = np.zeros((cap, 4), np.float32)
    ps = np.zeros(cap, np.float32)
    pl = np.zeros(cap, np.int64)
    plv = np.zeros(cap, np.int32)
    keep = np.zeros(cap, np.int64)
    n_pre = ctypes.c_int32(0)
    n_keep = ctypes.c_int32(0)
    FP = ctypes.POINTER(ctypes.c_float)
    hm_p = (FP * L)(*[_p(h, ctypes.c_float) for h in hm])
    reg_p = (FP * L)(*[_p(r, ctypes.c_float) for r in reg])
    rc = lib().oracle_decode_nms(ctypes.c_int(L), _p(H, ctypes.c_int32), _p(W, ctypes.c_int32), _p(S, ctypes.c_int32),
                                 hm_p, reg_p, ctypes.c_float(score_thresh), ctypes.c_int32(pre_topk),
                                 ctypes.c_float(nms_thresh), ctypes.c_int32(post_topk),
                                 _p(pb, ctypes.c_float), _p(ps, ctypes.c_float), _p(pl, ctypes.c_int64),
                                 _p(plv, ctypes.c_int32), ctypes.byref(n_pre), _p(keep, ctypes.c_int64),
                                 ctypes.byref(n_keep))
    assert rc == 0
    n, k = n_pre.value, n_keep.value
    keep = keep[:k].copy()
    return {"pre_boxes": pb[:n].copy(), "pre_scores": ps[:n].copy(), "pre_loc": pl[:n].copy(),
            "pre_level": plv[:n].copy(), "keep": keep, "boxes": pb[:n][keep], "scores": ps[:n][keep]}


def roi_predict(h, cls_w, cls_b, box_w, box_b, props, reg_weights, image_hw, score_thresh, nms_thresh, topk):
    h = np.ascontiguousarray(h, np.float32)
    n, Cc = h.shape
    arrs = [np.ascontiguousarray(a, np.float32) for a in (cls_w, cls_b, box_w, box_b, props)]
    rw = np.asarray(reg_weights, np.float32)
    cap = max(n, 1)
    raw_b, raw_s = np.zeros((cap, 4), np.float32), np.zeros(cap, np.float32)
    det_b, det_s = np.zeros((cap, 4), np.float32), np.zeros(cap, np.float32)
    det_src = np.zeros(cap, np.int64)
    cnt = ctypes.c_int32(0)
    rc = lib().oracle_roi_predict(_p(h, ctypes.c_float), ctypes.c_int64(n), ctypes.c_int32(Cc), _p(arrs[0], ctypes.c_float),
                                  _p(arrs[1], ctypes.c_float), _p(arrs[2], ctypes.c_float), _p(arrs[3], ctypes.c_float),
                                  _p(arrs[4], ctypes.c_float), _p(rw, ctypes.c_float), ctypes.c_float(image_hw[0]),
                                  ctypes.c_float(image_hw[1]), ctypes.c_float(score_thresh), ctypes.c_float(nms_thresh),
                                  ctypes.c_int32(topk), _p(raw_b, ctypes.c_float), _p(raw_s, ctypes.c_float), _p(det_b, ctypes.c_float),
                                  _p(det_s, ctypes.c_float), _p(det_src, ctypes.c_int64), ctypes.byref(cnt))
    assert rc == 0
    k = cnt.value
    return {"raw_boxes": raw_b[:n], "raw_scores": raw_s[:n], "boxes": det_b[:k].copy(), "scores": det_s[:k].copy(), "src": det_src[:k].copy()}


def roi_align_c(feat_chw: np.ndarray, boxes: np.ndarray, scale: float, pooled: int) -> np.ndarray:
    """ROIAlign (aligned, sampling_ratio 0) by oracle/ref_decode.c::oracle_roi_align -- the compiled twin of ref_model.roi_align's Python
    loop (same published algorithm, same fp32 operation order); used by bench.py's cpu_baseline and checked against the loop in tests."""
    f = np.ascontiguousarray(feat_chw, np.float32)
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
    Cc, H, W = f.shape
    out = np.zeros((len(b), Cc, pooled, pooled), np.float32)
    rc = lib().oracle_roi_align(_p(f, ctypes.c_float), ctypes.c_int32(Cc), ctypes.c_int32(H), ctypes.c_int32(W), _p(b, ctypes.c_float),
                                ctypes.c_int64(len(b)), ctypes.c_float(scale), ctypes.c_int32(pooled), _p(out, ctypes.c_float))
    assert rc == 0
    return out


# ----------------------------------------------------------------------------------------------
# Pure-numpy twin (small cases; mirrors ref_decode.c step for step, fp32 throughout).
# ----------------------------------------------------------------------------------------------
def nms_numpy(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    boxes = boxes.astype(np.float32).reshape(-1, 4)
    scores = scores.astype(np.float32)
    n = len(scores)
    order = sorted(range(n), key=lambda i: (-float(scores[i]), i))
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    dead = np.zeros(n, bool)
    keep: List[int] = []
    f0 = np.float32(0)
    for a in range(n):
        if dead[a]:
            continue
        i = order[a]
        keep.append(i)
        for c in range(a + 1, n):
            if dead[c]:
                continue
            j = order[c]
            w = max(f0, min(boxes[i, 2], boxes[j, 2]) - max(boxes[i, 0], boxes[j, 0]))
            h = max(f0, min(boxes[i, 3], boxes[j, 3]) - max(boxes[i, 1], boxes[j, 1]))
            inter = np.float32(w * h)
            with np.errstate(invalid="ignore", divide="ignore"):
                ovr = inter / np.float32(np.float32(area[i] + area[j]) - inter)
            if ovr > np.float32(thr):
                dead[c] = True
    return np.array(keep, dtype=np.int64)
