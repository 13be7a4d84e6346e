"""SURVEY.md section 8f rows 2-3 on the GPU: engine detections and oracle detections through the SAME evaluation entry
point (unina_yolo_dla_amd.evaluate = eval.py:18-138 + train.py:299-520) must give the same small-object TP/FP/FN and
the same conformal quantile; the lite_p2 table variant (model.py:184-190) against the fixture made by importing the
reference model.py."""
import json
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

SIZE = 640
SEEDS = (1234, 1235, 1236, 1237)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


@pytest.fixture(scope="module")
def dataset(pkg, oracle_mod, oracle_sd7, tmp_path_factory):
    """Frames + labels, a confidence threshold in the widest gap between oracle confidences around 0.5, and the number
    of oracle detections within the fp16 score tolerance of that threshold (near-threshold keep/drop flips are the
    comparison harness's business, SURVEY.md section 8c 'Tolerances': an engine whose scores move by up to 3.5e-3 may keep
    or drop exactly those)."""
    from evalset import OracleDetector, make_dataset
    det = OracleDetector(oracle_mod, oracle_sd7, pkg.graph.OUTPUT_NAMES)
    root = str(tmp_path_factory.mktemp("evalset"))
    make_dataset(root, pkg, det, SIZE, SEEDS, conf=0.5)
    confs = np.sort(np.concatenate([det(pkg.rng.frame(s, SIZE, SIZE)[0], 0.40, 0.45, 0.0)["confidence"] for s in SEEDS]))
    confs = confs[(confs > 0.45) & (confs < 0.55)]
    gaps = np.diff(confs)
    i = int(np.argmax(gaps))
    assert gaps[i] > 2e-4, gaps[i]                      # 100x the fp32 engine's score error
    thr = float((confs[i] + confs[i + 1]) / 2)
    from test_gpu_parity import FP16_SCORE_TOL
    borderline = int((np.abs(confs - thr) < FP16_SCORE_TOL).sum())
    return root, det, thr, borderline


@pytest.mark.parametrize("precision", ["fp16", "fp32"])
def test_engine_and_oracle_detections_give_the_same_metrics(pkg, sd7, dataset, torch_cuda, tmp_path, precision):
    from unina_yolo_dla_amd import evaluate as ev, export
    root, oracle_det, thr, borderline = dataset
    path = str(tmp_path / "m.une")
    export.export_engine(sd7, path, precision=export.FP32 if precision == "fp32" else export.FP16)
    det = ev.EngineDetector(path, autotune=False)
    try:
        got = ev.evaluate(det, root, SIZE, thr, 0.45, 0.0, str(tmp_path / "engine"), (det.width, det.height), 0.1)
    finally:
        det.close()
    want = ev.evaluate(oracle_det, root, SIZE, thr, 0.45, 0.0, str(tmp_path / "oracle"), None, 0.1)    # (q = 0: undilated boxes, some under 15 px)
    so_g, so_w = got["small_object"], want["small_object"]
    assert so_w["small_object_tp"] >= 8 and so_w["small_object_fn"] >= len(SEEDS) and so_w["small_object_fp"] > 0
    # fp32 engine: identical counts. fp16 engine: identical but for the detections whose oracle score lies within the
    # fp16 score tolerance of the threshold (each may be kept by one side and dropped by the other)
    slack = 0 if precision == "fp32" else borderline
    for k in ("small_object_tp", "small_object_fp", "small_object_fn"):
        assert abs(so_g[k] - so_w[k]) <= slack, (k, so_g, so_w, slack)
    assert abs(got["conformal"]["q_hat"] - want["conformal"]["q_hat"]) < 1e-3, (got["conformal"], want["conformal"])
    assert abs(got["conformal"]["num_calibration_samples"] - want["conformal"]["num_calibration_samples"]) <= (0 if precision == "fp32" else 2)
    recs = json.load(open(os.path.join(str(tmp_path / "engine"), "predictions.json")))
    assert abs(len(recs) - len(want["predictions"])) <= slack and set(recs[0]) == {"image_id", "category_id", "bbox", "score"}
    print(f"{precision}: engine {so_g} | oracle {so_w} | q_hat {got['conformal']['q_hat']:.6f} vs {want['conformal']['q_hat']:.6f} | borderline {borderline}")


def test_evaluate_camera_frames_through_the_stem_kernel(pkg, sd7, torch_cuda, tmp_path):
    """uint8 camera frames (any size) go through unina_infer_bgra; predictions.json boxes come back in the image's own
    pixels (eval.py:96-108 divides by the image's width / height)."""
    from unina_yolo_dla_amd import evaluate as ev, export
    path = str(tmp_path / "m.une")
    export.export_engine(sd7, path)
    root = tmp_path / "cam"
    (root / "images").mkdir(parents=True)
    (root / "labels").mkdir()
    rng = np.random.default_rng(3)
    np.save(root / "images" / "a.npy", rng.integers(0, 256, (360, 480, 3), dtype=np.uint8))
    (root / "labels" / "a.txt").write_text("0 0.5 0.5 0.02 0.02\n")
    det = ev.EngineDetector(path, autotune=False)
    try:
        res = ev.evaluate(det, str(root), 640, 0.3, 0.45, 0.1, None, (det.width, det.height))
    finally:
        det.close()
    assert res["images"] == 1 and res["small_object"]["small_object_fn"] + res["small_object"]["small_object_tp"] == 1
    for r in res["predictions"]:
        assert -60 < r["bbox"][0] < 480 + 60 and -60 < r["bbox"][1] < 360 + 60


def test_lite_p2_variant_vs_reference_fixture(pkg, oracle_mod, torch_cuda):
    """model.py:184-190 lite_p2=True (stage1's C3k2 replaced by one 3x3 ConvBlock) on the GPU against the heads the
    reference model.py itself produced (tests/golden/lite_p2_64_seed1234.npz), and detections against the oracle."""
    from unina_yolo_dla_amd.engine import Engine
    from test_gpu_parity import HEAD_ATOL
    gold = load_golden("lite_p2_64_seed1234.npz")
    g = pkg.graph.Graph(lite_p2=True, in_h=64, in_w=64)
    sd = pkg.synth.make_state_dict(7, g)
    e = Engine.from_state_dict(sd, g)
    try:
        x = pkg.rng.frame(1234, 64, 64)
        xd = torch_cuda.from_numpy(x).cuda()
        heads = {k: v.copy() for k, v in e.forward(xd).items()}
        for name in pkg.graph.OUTPUT_NAMES:
            np.testing.assert_allclose(heads[name], gold[f"head/{name}"], atol=HEAD_ATOL, rtol=0, err_msg=name)
        got = e.infer(xd, 0.05, 0.45, 0.1)
        want, ncand = oracle_mod.postprocess([gold[f"head/{n}"] for n in pkg.graph.OUTPUT_NAMES], 0.05, 0.45, 0.1)
        assert ncand > 10
        # 94 boxes at 64x64 and a 0.05 threshold: the hard fp16 bounds and the median hold; the share of boxes at
        # IoU >= 0.999 is 0.94 on this tiny sample (0.97-0.99 at 640x640)
        from detcmp import compare
        from test_gpu_parity import FP16_MIN_IOU, FP16_SCORE_TOL
        stats = compare(got, want, 0.05, min_iou=FP16_MIN_IOU, score_tol=FP16_SCORE_TOL)
        assert stats["matched"] >= 0.97 * len(want) and stats["median_dscore"] < 1e-3 and stats["frac_iou_ge_0.999"] >= 0.9, stats
        e.set_fusion(False)
        plain = e.forward(xd)
        from test_gpu_parity import same_head
        for k in plain:
            same_head(plain[k], heads[k], k)
    finally:
        e.close()
