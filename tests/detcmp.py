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


def compare(test, ref, conf_thr, min_iou=0.999, score_tol=1e-3, max_unmatched_frac=0.02, iou_thr=0.45, iou_margin=5e-3):
    """Returns a dict of statistics; raises AssertionError on a violation.

    Matched pairs must satisfy IoU >= min_iou and |score difference| < score_tol. A detection WITHOUT a partner must be
    an evidenced borderline keep/drop decision (SURVEY.md section 8c 'Tolerances'):
      (1) its score lies within score_tol of the confidence threshold, or
      (2) an NMS decision that provably hinges on a near-threshold quantity: a same-class box of the OTHER set overlaps it
          with an IoU within iou_margin of the NMS threshold `iou_thr` (the suppression flipped), or overlaps it above the
          threshold while (a) being unmatched itself (a borderline neighbour: the chain starts at a detection that is
          excused on its own) or (b) scoring within score_tol of it (the greedy order flipped).
    Anything else -- e.g. a box that merely sits near some other box -- is a real divergence and fails."""
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

    def excused(d, overlaps, other, other_unmatched):
        """d: the unmatched record; overlaps: its same-class IoUs with every record of `other`"""
        if abs(float(d["confidence"]) - conf_thr) <= score_tol:
            return True
        if len(other) == 0:
            return False
        if (np.abs(overlaps - iou_thr) <= iou_margin).any():
            return True
        above = overlaps > iou_thr
        if (above & other_unmatched).any():
            return True
        return bool((above & (np.abs(other["confidence"] - d["confidence"]) <= score_tol)).any())

    ref_unmatched = ~used_ref
    test_unmatched = np.ones(len(test), bool)
    test_unmatched[ti] = False
    for i in un_t:
        assert excused(test[i], m[i], ref, ref_unmatched), \
            f"test detection {i} {test[i]} has no reference partner and no borderline decision explains it"
    for j in un_r:
        assert excused(ref[j], m[:, j], test, test_unmatched), \
            f"reference detection {j} {ref[j]} was not reproduced and no borderline decision explains it"
    total = max(len(ref), 1)
    assert (len(un_t) + len(un_r)) / total <= max_unmatched_frac, (len(un_t), len(un_r), total)
    ious = m[ti, rj] if len(pairs) else np.ones(1)
    return dict(matched=len(pairs), unmatched_test=len(un_t), unmatched_ref=len(un_r),
                min_iou=float(ious.min()), max_dscore=float(dscore.max()) if len(pairs) else 0.0,
                median_dscore=float(np.median(dscore)) if len(pairs) else 0.0,
                **{"frac_iou_ge_0.999": float((ious >= 0.999).mean())})
