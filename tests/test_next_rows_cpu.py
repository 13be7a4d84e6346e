"""SURVEY.md section 8f rows on CPU: SmallObjectMetric pinned by the reference's only known-answer vector
(data_loader.py:427-436), eval conversions, conformal calibration, and the pre-process oracle's arithmetic."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD


def _dets(rows):
    from unina_yolo_dla_amd.gather import DET_DTYPE
    d = np.zeros(len(rows), dtype=DET_DTYPE)
    for i, (x1, y1, x2, y2, conf, cls) in enumerate(rows):
        d[i] = (x1, y1, x2, y2, conf, cls, 1, 0)
    return d


def test_small_object_metric_known_answer(pkg):
    ex = json.load(open(os.path.join(GOLD, "small_object_metric_example.json")))
    m = pkg.metrics.SmallObjectMetric(ex["small_threshold_px"], ex["iou_threshold"], ex["image_size"])
    m.update([np.array(ex["preds"])], [np.array(ex["targets"])])
    r = m.compute()
    assert (r["small_object_tp"], r["small_object_fp"], r["small_object_fn"]) == (0, 1, 1)
    assert m._iou(ex["preds"][0][:4], ex["targets"][0][1:5]) == pytest.approx(ex["expected"]["iou"], abs=2e-4)
    # a matching small prediction flips it to TP=1
    m.reset()
    m.update([np.array([[0.5, 0.5, 0.0105, 0.0205, 0.9, 0]])], [np.array(ex["targets"])])
    assert m.compute()["small_object_tp"] == 1 and m.compute()["small_object_f1"] == 1.0
    # class mismatch / large GT ignored / empty predictions
    m.reset()
    m.update([np.zeros((0, 6))], [np.array(ex["targets"])])
    assert m.compute()["small_object_fn"] == 1


def test_eval_conversions_roundtrip(pkg):
    d = _dets([(100, 200, 110, 214, 0.9, 2)])
    rec = pkg.metrics.detections_to_coco(d, "img0")
    assert rec[0]["bbox"] == [100.0, 200.0, 10.0, 14.0] and rec[0]["category_id"] == 2      # eval.py:58-61
    rows = pkg.metrics.coco_to_metric_rows(rec, 640, 640)
    np.testing.assert_allclose(rows[0], [105 / 640, 207 / 640, 10 / 640, 14 / 640, 0.9, 2], rtol=1e-6)   # eval.py:96-108
    labels = [np.array([[2, 105 / 640, 207 / 640, 10 / 640, 14 / 640]])]
    r = pkg.metrics.evaluate_small_objects([d], labels)
    assert r["small_object_tp"] == 1 and r["small_object_fp"] == 0


def test_conformal_quantile(pkg):
    # three images, one GT each; predictions with IoU 1.0, 0.81, 0.64 -> scores 0, 0.19, 0.36
    labels = [np.array([[0, 0.5, 0.5, 0.1, 0.1]])] * 3
    dets = [_dets([(288, 288, 352, 352, 0.9, 0)]), _dets([(288, 288, 352, 352 - 12.16, 0.8, 0)]),
            _dets([(288, 288, 352 - 12.8, 352 - 12.8, 0.7, 0), (0, 0, 10, 10, 0.95, 0)])]
    r = pkg.metrics.conformal_quantile(dets, labels, alpha=0.1)
    assert r["num_calibration_samples"] == 3
    assert r["q_hat"] == pytest.approx(float(np.quantile([0.0, 0.19, 0.36], 0.9)), abs=1e-3)   # train.py:499
    with pytest.raises(ValueError):
        pkg.metrics.conformal_quantile([_dets([(0, 0, 5, 5, 0.9, 1)])], labels[:1])


def test_preprocess_oracle_arithmetic(oracle_mod):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (6, 8, 4), dtype=np.uint8)
    out = oracle_mod.preprocess_bgra(img)
    mean, std = np.array(oracle_mod.IMAGENET[:3], np.float32), np.array(oracle_mod.IMAGENET[3:], np.float32)
    want = ((img[..., [2, 1, 0]].astype(np.float32) / np.float32(255)) - mean) / std          # BGR -> RGB, :118-127
    np.testing.assert_array_equal(out, want.transpose(2, 0, 1))
    # identity-size "resize" reproduces the plain conversion up to rounding of the bilinear weights (w00 = 1)
    np.testing.assert_allclose(oracle_mod.preprocess_bgra(img, dst_hw=(6, 8)), out, atol=1e-6)
    # 2x downscale of a constant image is that constant
    flat = np.full((8, 8, 4), 77, np.uint8)
    np.testing.assert_allclose(oracle_mod.preprocess_bgra(flat, dst_hw=(4, 4)), oracle_mod.preprocess_bgra(flat)[:, :4, :4], atol=1e-6)
    # NV12: grey (U=V=128) gives R=G=B=Y
    y = rng.integers(0, 256, (4, 6), dtype=np.uint8)
    uv = np.full((2, 6), 128, np.uint8)
    out = oracle_mod.preprocess_nv12(y, uv)
    want = ((y.astype(np.float32) / np.float32(255))[None] - mean[:, None, None]) / std[:, None, None]
    np.testing.assert_array_equal(out, want)


def test_evaluate_entry_point_on_cpu(pkg, oracle_mod, oracle_sd7, tmp_path):
    """unina_yolo_dla_amd.evaluate.evaluate (eval.py:18-138 + train.py:299-520) end to end with the ORACLE as the
    detector (no GPU): predictions.json schema, per-image grouping, small-object counts, conformal quantile."""
    import json
    from evalset import OracleDetector, make_dataset
    from unina_yolo_dla_amd import evaluate as ev
    size = 128
    det = OracleDetector(oracle_mod, oracle_sd7, pkg.graph.OUTPUT_NAMES)
    root = str(tmp_path / "ds")
    n_labels = make_dataset(root, pkg, det, size, (1234, 1235), conf=0.3)
    out = str(tmp_path / "run")
    res = ev.evaluate(det, root, imgsz=size, conf=0.3, iou=0.45, conformal_q=0.0, out_dir=out, conformal_alpha=0.1)
    recs = json.load(open(os.path.join(out, "predictions.json")))
    assert res["images"] == 2 and len(recs) == len(res["predictions"]) > 4
    assert set(recs[0]) == {"image_id", "category_id", "bbox", "score"} and recs[0]["image_id"] == "frame1234"   # eval.py:58-61
    assert all(len(r["bbox"]) == 4 and r["bbox"][2] > 0 for r in recs)
    so = res["small_object"]
    assert so["small_object_tp"] > 0 and so["small_object_fn"] >= 2          # the jittered labels match; the planted small GTs do not
    # same numbers through the lower-level helper (eval.py:110-131 loop)
    dets = [det(np.load(f), 0.3, 0.45, 0.0) for f in ev.list_frames(root)]
    labels = [ev.read_labels(ev.label_path(f)) for f in ev.list_frames(root)]
    assert sum(len(l) for l in labels) == n_labels
    direct = pkg.metrics.evaluate_small_objects(dets, labels, size, size)
    assert (direct["small_object_tp"], direct["small_object_fp"], direct["small_object_fn"]) == \
        (so["small_object_tp"], so["small_object_fp"], so["small_object_fn"])
    c = res["conformal"]
    assert 0.0 < c["q_hat"] < 0.5 and c["num_calibration_samples"] > 4
    assert json.load(open(os.path.join(out, "conformal_params.json")))["q_hat"] == c["q_hat"]
    with pytest.raises(FileNotFoundError):
        ev.evaluate(det, str(tmp_path / "empty"))
