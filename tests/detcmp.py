"""Detection-set comparison with the north star's tolerance (BASELINE.json): matched boxes IoU >= 0.999 and
|score difference| < 1e-3; detections without a partner must be explained by a borderline decision
(score within `tol` of the confidence threshold, or an NMS decision that hinges on a near-threshold IoU /
a borderline neighbour)."""
import numpy as np


def iou_matrix(a, b):
    ix1 = np.maximum(a["x1"][:, None], b["x1"][None, :])
    iy1 = np.maximum(a["y1"][:, None], b["y1"][None, :])
    ix2 = np.minimum(a["x2"][:, None], b["x2"][None, :])
    iy2 = np.minimum(a["y2"][:, None], b["y2"][None, :])
    inter = np.clip(ix2 - ix1, 0, None) * np.clip(iy2 - iy1, 0, None)
    aa = (a["x2"] - a["x1"]) * (a["y2"] - a["y1"])
    ab = (b["x2"] - b["x1"]) * (b["y2"] - b["y1"])
    return inter / np.maximum(aa[:, None] + ab[None, :] - inter, 1e-12)


def compare(test, ref, conf_thr, min_iou=0.999, score_tol=1e-3, max_unmatched_frac=0.02):
    """Returns a dict of statistics; raises AssertionError on a violation."""
    if len(test) == 0 and len(ref) == 0:
        return {"matched": 0, "unmatched_test": 0, "unmatched_ref": 0, "min_iou": 1.0, "max_dscore": 0.0,
                "median_dscore": 0.0, "frac_iou_ge_0.999": 1.0}
    m = iou_matrix(test, ref) if len(test) and len(ref) else np.zeros((len(test), len(ref)))
    same = test["class_id"][:, None] == ref["class_id"][None, :] if len(test) and len(ref) else m.astype(bool)
    m = np.where(same, m, 0.0)
    used_ref = np.zeros(len(ref), bool)
    pairs = []
    for i in np.argsort(-test["confidence"]):
        if not len(ref):
            break
        j = int(np.argmax(np.where(used_ref, -1.0, m[i])))
        if not used_ref[j] and m[i, j] >= min_iou:
            used_ref[j] = True
            pairs.append((i, j))
    ti = np.array([p[0] for p in pairs], int)
    rj = np.array([p[1] for p in pairs], int)
    dscore = np.abs(test["confidence"][ti] - ref["confidence"][rj]) if len(pairs) else np.zeros(0)
    assert (dscore < score_tol).all(), f"score drift {dscore.max():.3e} >= {score_tol}"
    un_t = np.setdiff1d(np.arange(len(test)), ti)
    un_r = np.setdiff1d(np.arange(len(ref)), rj)
    # an unmatched detection must be a borderline keep/drop: near the confidence threshold, or involved in an NMS
    # decision (it overlaps a same-class detection of the other set strongly enough to have been suppressed there)
    def excused(d, other):
        if abs(float(d["confidence"]) - conf_thr) <= score_tol:
            return True
        if len(other) == 0:
            return False
        one = np.array([d], dtype=other.dtype) if d.dtype == other.dtype else None
        o = iou_matrix(one if one is not None else np.array([tuple(d[n] for n in other.dtype.names)], dtype=other.dtype), other)[0]
        o = np.where(other["class_id"] == d["class_id"], o, 0.0)
        return bool(o.max() > 0.3)
    for i in un_t:
        assert excused(test[i], ref), f"test detection {i} {test[i]} has no reference partner and is not borderline"
    for j in un_r:
        assert excused(ref[j], test), f"reference detection {j} {ref[j]} was not reproduced and is not borderline"
    total = max(len(ref), 1)
    assert (len(un_t) + len(un_r)) / total <= max_unmatched_frac, (len(un_t), len(un_r), total)
    ious = m[ti, rj] if len(pairs) else np.ones(1)
    return dict(matched=len(pairs), unmatched_test=len(un_t), unmatched_ref=len(un_r),
                min_iou=float(ious.min()), max_dscore=float(dscore.max()) if len(pairs) else 0.0,
                median_dscore=float(np.median(dscore)) if len(pairs) else 0.0,
                **{"frac_iou_ge_0.999": float((ious >= 0.999).mean())})
