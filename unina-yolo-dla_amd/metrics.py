"""Evaluation-side mirror of the reference, numpy only:

* ``SmallObjectMetric``            -- data_loader.py:249-414 (P/R/F1 on ground-truth boxes < 15 px)
* ``detections_to_coco`` / ``coco_to_metric_rows`` -- the predictions.json schema eval.py reads (eval.py:58-61) and its
  xyxy-pixels -> normalised-xywh conversion (eval.py:96-108)
* ``evaluate_small_objects``       -- eval.py:110-131's loop over images, fed by engine detections
* ``conformal_quantile``           -- train.py:299-520: greedy IoU>=0.5 matching, scores 1-IoU, q_hat = quantile(1-alpha);
  q_hat is what the engine takes as ``conformal_q`` (perception_node.cpp:389)
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence

import numpy as np


class SmallObjectMetric:
    """data_loader.py:249-414. predictions: per image [N,6] = xc,yc,w,h (normalised), conf, cls;
    ground truths: per image [M,5] = cls, xc,yc,w,h (normalised)."""

    def __init__(self, size_threshold: int = 15, iou_threshold: float = 0.5, image_size: int = 640):
        self.size_threshold, self.iou_threshold, self.image_size = size_threshold, iou_threshold, image_size
        self.reset()

    def reset(self) -> None:
        self.true_positives = self.false_positives = self.false_negatives = 0

    def _is_small(self, w_norm: float, h_norm: float) -> bool:          # data_loader.py:285-290
        return w_norm * self.image_size < self.size_threshold and h_norm * self.image_size < self.size_threshold

    @staticmethod
    def _iou(b1, b2) -> float:                                          # data_loader.py:292-321 (xywh, centre)
        ax1, ay1, ax2, ay2 = b1[0] - b1[2] / 2, b1[1] - b1[3] / 2, b1[0] + b1[2] / 2, b1[1] + b1[3] / 2
        bx1, by1, bx2, by2 = b2[0] - b2[2] / 2, b2[1] - b2[3] / 2, b2[0] + b2[2] / 2, b2[1] + b2[3] / 2
        inter = max(0.0, min(ax2, bx2) - max(ax1, bx1)) * max(0.0, min(ay2, by2) - max(ay1, by1))
        union = (ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter
        return 0.0 if union <= 0 else float(inter / union)

    def update(self, predictions: Sequence[np.ndarray], ground_truths: Sequence[np.ndarray]) -> None:
        for preds, gts in zip(predictions, ground_truths):
            preds = np.asarray(preds, dtype=np.float64).reshape(-1, 6)
            gts = np.asarray(gts, dtype=np.float64).reshape(-1, 5)
            small = [g for g in gts if self._is_small(g[3], g[4])]
            if not small:
                continue
            if len(preds) == 0:
                self.false_negatives += len(small)
                continue
            matched = set()
            for pred in preds[np.argsort(-preds[:, 4], kind="stable")]:
                best_iou, best = 0.0, -1
                for i, gt in enumerate(small):
                    if i in matched or int(pred[5]) != int(gt[0]):
                        continue
                    iou = self._iou(pred[:4], gt[1:5])
                    if iou > best_iou:
                        best_iou, best = iou, i
                if best_iou >= self.iou_threshold:
                    self.true_positives += 1
                    matched.add(best)
                elif self._is_small(pred[2], pred[3]):                  # FP only if the prediction is itself small
                    self.false_positives += 1
            self.false_negatives += len(small) - len(matched)

    def compute(self) -> Dict[str, float]:
        tp, fp, fn = self.true_positives, self.false_positives, self.false_negatives
        precision = tp / (tp + fp) if tp + fp else 0.0
        recall = tp / (tp + fn) if tp + fn else 0.0
        f1 = 2 * precision * recall / (precision + recall) if precision + recall else 0.0
        return {"small_object_precision": precision, "small_object_recall": recall, "small_object_f1": f1,
                "small_object_tp": tp, "small_object_fp": fp, "small_object_fn": fn}


def detections_to_coco(dets: np.ndarray, image_id: str) -> List[dict]:
    """Engine detections (xyxy pixels) -> predictions.json records (eval.py:58-61)."""
    return [{"image_id": image_id, "category_id": int(d["class_id"]),
             "bbox": [float(d["x1"]), float(d["y1"]), float(d["x2"] - d["x1"]), float(d["y2"] - d["y1"])],
             "score": float(d["confidence"])} for d in dets]


def coco_to_metric_rows(records: Iterable[dict], width: int, height: int) -> np.ndarray:
    """predictions.json records -> [N,6] xc,yc,w,h (normalised), conf, cls (eval.py:96-108)."""
    rows = [[(r["bbox"][0] + r["bbox"][2] / 2) / width, (r["bbox"][1] + r["bbox"][3] / 2) / height,
             r["bbox"][2] / width, r["bbox"][3] / height, r["score"], r["category_id"]] for r in records]
    return np.asarray(rows, dtype=np.float64).reshape(-1, 6)


def evaluate_small_objects(dets_per_image: Sequence[np.ndarray], labels_per_image: Sequence[np.ndarray],
                           width: int = 640, height: int = 640, size_threshold: int = 15) -> Dict[str, float]:
    """eval.py:110-131: per image, detections -> normalised rows, labels = YOLO txt rows [cls,xc,yc,w,h]."""
    metric = SmallObjectMetric(size_threshold=size_threshold, image_size=width)
    for i, (dets, labels) in enumerate(zip(dets_per_image, labels_per_image)):
        rows = coco_to_metric_rows(detections_to_coco(dets, str(i)), width, height)
        metric.update([rows], [np.asarray(labels, dtype=np.float64).reshape(-1, 5)])
    return metric.compute()


def _box_iou_xyxy(a, b) -> float:                                       # train.py:330-344
    x1, y1, x2, y2 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
    if x2 <= x1 or y2 <= y1:
        return 0.0
    inter = (x2 - x1) * (y2 - y1)
    union = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
    return inter / union if union > 0 else 0.0


def conformal_quantile(dets_per_image: Sequence[np.ndarray], labels_per_image: Sequence[np.ndarray], alpha: float = 0.10,
                       imgsz: int = 640) -> Dict[str, float]:
    """train.py:299-520. Detections: engine records (xyxy pixels, run with a very low confidence threshold, :403);
    labels: YOLO rows [cls,xc,yc,w,h] normalised (:346-352 converts with imgsz). Greedy matching by confidence, same
    class, best IoU >= 0.5 (:455-482); score = 1 - IoU (:486); q_hat = np.quantile(scores, 1-alpha) (:499)."""
    scores: List[float] = []
    for dets, labels in zip(dets_per_image, labels_per_image):
        labels = np.asarray(labels, dtype=np.float64).reshape(-1, 5)
        gts = [((l[1] - l[3] / 2) * imgsz, (l[2] - l[4] / 2) * imgsz, (l[1] + l[3] / 2) * imgsz, (l[2] + l[4] / 2) * imgsz)
               for l in labels]
        matched = set()
        for d in dets[np.argsort(-dets["confidence"], kind="stable")]:
            box = (d["x1"], d["y1"], d["x2"], d["y2"])
            best_iou, best = 0.0, -1
            for gi, (g, l) in enumerate(zip(gts, labels)):
                if gi in matched or int(l[0]) != int(d["class_id"]):
                    continue
                iou = _box_iou_xyxy(box, g)
                if iou > best_iou and iou >= 0.5:
                    best_iou, best = iou, gi
            if best >= 0:
                matched.add(best)
                scores.append(1.0 - best_iou)
    if not scores:
        raise ValueError("conformal calibration failed: no matched predictions")          # train.py:492-496
    s = np.asarray(scores)
    q = float(np.quantile(s, 1 - alpha))
    return {"alpha": alpha, "coverage_target": 1 - alpha, "q_hat": q, "dilation_factor": q,
            "num_calibration_samples": len(scores), "mean_nonconformity": float(s.mean()),
            "std_nonconformity": float(s.std())}
