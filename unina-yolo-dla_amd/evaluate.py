"""Evaluation entry point over the engine: the counterpart of the reference's ``eval.py::evaluate_model``
(eval.py:18-138) and of ``train.py::calibrate_conformal_prediction`` (train.py:299-520) -- SURVEY.md section 8f rows 2-3.

The reference runs ``ultralytics.YOLO(weights).val(save_json=True)``, re-reads ``predictions.json`` (records
``{'image_id': stem, 'category_id': int, 'bbox': [x_min, y_min, w, h] px, 'score': float}``, eval.py:58-61), walks the
validation images of a YOLO-layout dataset (``<root>/images/<stem>.*`` beside ``<root>/labels/<stem>.txt``, rows
``cls xc yc w h`` normalised, eval.py:110-121), converts each prediction with the image's own width / height
(eval.py:96-108) and feeds ``SmallObjectMetric`` (data_loader.py:249-414). Here the detector is an engine file run
through the C ABI (``unina_infer`` / ``unina_infer_bgra``); everything after the detector is the same sequence.

    python -m unina_yolo_dla_amd.evaluate --engine model.une --data <root> [--out-dir runs/eval] [--conformal]

Frames: ``.npy`` holding the network tensor itself (fp32 ``[3,H,W]``, H x W = the engine's input size) or a camera
frame (uint8 ``[h,w,4]`` BGRA / ``[h,w,3]`` RGB, any size: resized + normalised on the GPU by the stem kernel, i.e.
``unina_infer_bgra``), and ``.png`` / ``.jpg`` when Pillow is importable. Detections come back in network pixels; the
records written to predictions.json are scaled to the image's own pixels, as Ultralytics writes them.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import metrics

IMAGE_EXT = (".npy", ".png", ".jpg", ".jpeg")


def list_frames(root: str) -> List[str]:
    """Validation images of a YOLO-layout dataset: <root>/images/* (or <root>/* when there is no images/ dir)."""
    d = os.path.join(root, "images") if os.path.isdir(os.path.join(root, "images")) else root
    files = [f for f in sorted(glob.glob(os.path.join(d, "*"))) if f.lower().endswith(IMAGE_EXT)]
    return files


def label_path(frame_path: str) -> str:
    """eval.py:110-112: labels live in the sibling 'labels' directory, <stem>.txt."""
    stem = os.path.splitext(os.path.basename(frame_path))[0]
    return os.path.join(os.path.dirname(os.path.dirname(frame_path)), "labels", stem + ".txt")


def read_labels(path: str) -> np.ndarray:
    """YOLO txt rows 'cls xc yc w h' (normalised) -> [M,5] (eval.py:113-121); a missing file is an empty image."""
    rows = []
    if os.path.exists(path):
        with open(path) as f:
            for line in f:
                p = line.split()
                if len(p) >= 5:
                    rows.append([float(p[0])] + [float(v) for v in p[1:5]])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 5)


def load_frame(path: str) -> np.ndarray:
    if path.lower().endswith(".npy"):
        return np.load(path, allow_pickle=False)
    try:
        from PIL import Image
    except ImportError as e:                                         # pragma: no cover
        raise RuntimeError(f"{path}: reading image files needs Pillow; store frames as .npy instead") from e
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)


def frame_size(frame: np.ndarray) -> Tuple[int, int]:
    """(width, height) of the image the labels are normalised to."""
    if frame.dtype == np.uint8:
        return frame.shape[1], frame.shape[0]
    return frame.shape[-1], frame.shape[-2]


class EngineDetector:
    """frame -> detections through the C ABI. fp32 [3,H,W] frames go to unina_infer; uint8 camera frames are packed to
    BGRA and go to unina_infer_bgra (pre-process inside the stem kernel, perception_node.cpp:601-656 as one call)."""

    def __init__(self, engine_path: str, device: int = 0, autotune: bool = True):
        from .engine import Engine
        self.eng = Engine(engine_path, device)
        if autotune:
            self.eng.autotune()
        self.width, self.height = self.eng.width, self.eng.height

    def __call__(self, frame: np.ndarray, conf: float, iou: float, q: float) -> np.ndarray:
        import torch
        dev = torch.device("cuda", self.eng.device)
        if frame.dtype == np.uint8:
            h, w = frame.shape[:2]
            if frame.shape[2] == 3:                                  # RGB -> BGRA
                bgra = np.empty((h, w, 4), dtype=np.uint8)
                bgra[..., 0], bgra[..., 1], bgra[..., 2], bgra[..., 3] = frame[..., 2], frame[..., 1], frame[..., 0], 255
            else:
                bgra = np.ascontiguousarray(frame)
            cam = torch.from_numpy(bgra.reshape(h, w * 4)).to(dev)
            return self.eng.infer_bgra(cam, w, h, w * 4, None, conf, iou, q)
        x = np.ascontiguousarray(frame, dtype=np.float32).reshape(1, 3, self.height, self.width)
        return self.eng.infer(torch.from_numpy(x).to(dev), conf, iou, q)

    def close(self):
        self.eng.close()


def evaluate(detect: Callable[[np.ndarray, float, float, float], np.ndarray], root: str, imgsz: int = 640,
             conf: float = 0.5, iou: float = 0.45, conformal_q: float = 0.1, out_dir: Optional[str] = None,
             net_size: Optional[Tuple[int, int]] = None, conformal_alpha: Optional[float] = None,
             conformal_conf: float = 0.001) -> Dict[str, object]:
    """eval.py:18-138 over `detect` (frame, conf, iou, q) -> GpuDetection records in network pixels.

    Writes <out_dir>/predictions.json (eval.py:58-61 schema, boxes in the image's own pixels), computes
    SmallObjectMetric(size_threshold=15, image_size=imgsz) exactly as eval.py:74,96-131 does, and -- with
    `conformal_alpha` -- the conformal quantile of train.py:299-520 from a second pass at a very low confidence
    threshold (train.py:403) without dilation. Returns {'small_object': {...}, 'conformal': {...} | None,
    'predictions': [...], 'images': n}."""
    files = list_frames(root)
    if not files:
        raise FileNotFoundError(f"no frames (*.npy / *.png / *.jpg) under {root}")
    records: List[dict] = []
    so = metrics.SmallObjectMetric(size_threshold=15, image_size=imgsz)           # eval.py:74
    conf_dets, conf_labels = [], []
    for path in files:
        stem = os.path.splitext(os.path.basename(path))[0]
        frame = load_frame(path)
        w, h = frame_size(frame)
        if frame.dtype == np.uint8:                                   # camera frame: detections come back in network pixels
            if net_size is None:
                raise ValueError("uint8 camera frames need net_size=(width, height) of the engine input")
            nw, nh = net_size
        else:                                                         # the network tensor itself
            nw, nh = w, h
        sx, sy = w / nw, h / nh
        dets = detect(frame, conf, iou, conformal_q)
        scaled = dets.copy()
        scaled["x1"], scaled["x2"] = dets["x1"] * sx, dets["x2"] * sx
        scaled["y1"], scaled["y2"] = dets["y1"] * sy, dets["y2"] * sy
        recs = metrics.detections_to_coco(scaled, stem)
        records += recs
        labels = read_labels(label_path(path))
        so.update([metrics.coco_to_metric_rows(recs, w, h)], [labels])           # eval.py:96-108, 124
        if conformal_alpha is not None:
            d0 = detect(frame, conformal_conf, iou, 0.0)
            # conformal_quantile converts labels with ONE size (train.py:346-352): hand it boxes in imgsz pixels
            c = d0.copy()
            c["x1"], c["x2"] = d0["x1"] * (imgsz / nw), d0["x2"] * (imgsz / nw)
            c["y1"], c["y2"] = d0["y1"] * (imgsz / nh), d0["y2"] * (imgsz / nh)
            conf_dets.append(c)
            conf_labels.append(labels)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "predictions.json"), "w") as f:
            json.dump(records, f)
    result: Dict[str, object] = {"images": len(files), "predictions": records, "small_object": so.compute(), "conformal": None}
    if conformal_alpha is not None:
        result["conformal"] = metrics.conformal_quantile(conf_dets, conf_labels, conformal_alpha, imgsz)
        if out_dir:
            with open(os.path.join(out_dir, "conformal_params.json"), "w") as f:     # train.py:506-520 writes the same keys
                json.dump(result["conformal"], f, indent=2)
    return result


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(description="small-object P/R/F1 (+ conformal quantile) of an engine file over a YOLO-layout dataset")
    ap.add_argument("--engine", required=True, help=".une engine file (export.export_engine)")
    ap.add_argument("--data", required=True, help="dataset root: images/ + labels/")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--conf", type=float, default=0.5)
    ap.add_argument("--iou", type=float, default=0.45)
    ap.add_argument("--conformal-q", type=float, default=0.1)
    ap.add_argument("--conformal", action="store_true", help="also calibrate q_hat (train.py:299-520)")
    ap.add_argument("--alpha", type=float, default=0.10)
    ap.add_argument("--out-dir", default="runs/eval")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    det = EngineDetector(a.engine, a.device)
    try:
        res = evaluate(det, a.data, a.imgsz, a.conf, a.iou, a.conformal_q, a.out_dir, (det.width, det.height),
                       a.alpha if a.conformal else None)
    finally:
        det.close()
    print(f"{res['images']} images, {len(res['predictions'])} predictions -> {os.path.join(a.out_dir, 'predictions.json')}")
    for k, v in res["small_object"].items():                                      # eval.py:128-131
        print(f"    {k}: {v}")
    if res["conformal"]:
        print(f"    q_hat: {res['conformal']['q_hat']:.6f}  ({res['conformal']['num_calibration_samples']} matched boxes)")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
