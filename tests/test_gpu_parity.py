"""GPU parity tests proper (-m gpu): the HIP engine, called through the C ABI, against the CPU oracle and the
committed reference fixtures. Post-process on identical inputs is exact; the forward is held to BASELINE.json's
tolerance (matched boxes IoU >= 0.999, |score delta| < 1e-3 vs fp32) as far as the fp16 format allows -- see the
tolerance block below."""
import numpy as np
import pytest

from conftest import load_golden
from detcmp import compare

pytestmark = pytest.mark.gpu

# ---- fp16 tolerances -------------------------------------------------------------------------------------
# North star (BASELINE.json): every matched box IoU >= 0.999 and |dscore| < 1e-3 against the fp32 forward.
# The fp32 engine (native fp32 MFMA) meets that outright on every detection (test_fp32_engine_meets_north_star_tolerance).
# The fp16 engine CANNOT, and that is the storage format's floor, not a kernel property -- proven on the CPU by
# tools/fp16_error_budget.py (committed run: profiles/r02/fp16_error_budget.txt): every fp16 rounding (folded weights,
# activation stores) of every layer group contributes a few percent of the head-error variance, no group dominates,
# and even with every layer but stem+stage1 kept in fp32 the worst of 4 711 detections still moves by 1.8e-3. With all
# roundings on, a bit-level emulation of the arithmetic (tests/emulate.py) gives, over 10 frames / 4 711 detections:
# max |dscore| 3.0e-3, p99 2.1e-3, min IoU 0.99858, 1.5 % of boxes below 0.999 -- the floor of ANY engine that stores
# fp16 weights and activations for this graph. So the fp16 engine is held to:
#   (a) the floor itself: its worst detection may not be worse than the emulation's on the same frames (+30 %)
#       -> test_fp16_engine_tail_is_the_format_floor;
#   (b) the north-star numbers on the typical detection: median |dscore| < 1e-3, >= 95 % of boxes at IoU >= 0.999;
#   (c) hard bounds just above the measured floor on every detection: IoU >= 0.9983, |dscore| < 3.5e-3.
HEAD_ATOL = 2.5e-2          # max |logit error| (cls std 2.0); measured 1.4e-2
CLS_RMS, REG_RMS = 6e-3, 1.5e-3   # measured 3.9e-3 / 7.3e-4
FP16_MIN_IOU, FP16_SCORE_TOL = 0.9983, 3.5e-3


def check_fp16_detections(got, want, conf_thr):
    stats = compare(got, want, conf_thr, min_iou=FP16_MIN_IOU, score_tol=FP16_SCORE_TOL)
    assert stats["matched"] >= 0.97 * len(want), stats
    assert stats["median_dscore"] < 1e-3, stats              # north-star score tolerance on the typical detection
    assert stats["frac_iou_ge_0.999"] >= 0.95, stats          # north-star box tolerance on >= 95 % of detections
    return stats


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


@pytest.fixture(scope="module")
def eng640(pkg, sd7, torch_cuda):
    from unina_yolo_dla_amd.engine import Engine
    e = Engine.from_state_dict(sd7)
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng64(pkg, sd7, torch_cuda):
    from unina_yolo_dla_amd.engine import Engine
    e = Engine.from_state_dict(sd7, pkg.graph.Graph(in_h=64, in_w=64))
    yield e
    e.close()


def _frame(pkg, torch, seed, s):
    return torch.from_numpy(pkg.rng.frame(seed, s, s)).cuda()


def test_library_is_the_hip_engine(eng64):
    """The loaded binary is the HIP engine AND was built from the sources under test: unina_version() carries the hash of
    csrc/ + include/ at build time (the .so travels to the GPU box as a file; nothing else ties it to the tree)."""
    from unina_yolo_dla_amd import build
    v = eng64.L.unina_version()
    assert b"gfx950" in v
    assert v.endswith(b"src:" + build.source_hash().encode()), (v, build.source_hash())
    assert len(eng64.op_infos()) == 52


@pytest.fixture
def unfused64(eng64):
    """Per-layer checks read the C3k2 blocks' internal buffers, which only the unfused op table writes."""
    eng64.set_fusion(False)
    yield eng64
    assert eng64.set_fusion(True) == 9


def test_mini64_every_buffer_vs_oracle(pkg, unfused64, oracle_mod, oracle_sd7, torch_cuda):
    eng64 = unfused64
    x = pkg.rng.frame(1234, 64, 64)
    heads = eng64.forward(torch_cuda.from_numpy(x).cuda())
    ref = oracle_mod.forward(oracle_sd7, x, keep_all=True)
    for name in pkg.graph.OUTPUT_NAMES:
        np.testing.assert_allclose(heads[name], ref[name], atol=HEAD_ATOL, rtol=0, err_msg=name)
    # intermediate buffers (engine buffer name -> oracle tensor; concat buffers compare against the oracle's cat)
    pairs = {
        "backbone.stem": "backbone.stem", "backbone.stage1_conv": "backbone.stage1_conv",
        "backbone.stage1_block.cat": "backbone.stage1_block.cat", "backbone.stage2_conv": "backbone.stage2_conv",
        "backbone.stage3_conv": "backbone.stage3_conv", "backbone.sppf.cat": "backbone.sppf.cat",
        "backbone.sppf": "backbone.sppf.cv2", "neck.cat_fpn1": "neck.cat_fpn1", "neck.cat_fpn2": "neck.cat_fpn2",
        "neck.cat_pan1": "neck.cat_pan1", "neck.cat_pan2": "neck.cat_pan2", "p2_fused": "neck.fpn_c3k2_2.cv3",
        "p3_out": "neck.pan_c3k2_1.cv3", "p4_out": "neck.pan_c3k2_2.cv3",
    }
    for bname, oname in pairs.items():
        got = eng64.read_buffer(bname)
        want = ref[oname]
        assert got.shape == want.shape, (bname, got.shape, want.shape)
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got, want, atol=1e-2 * scale, rtol=0, err_msg=bname)


def test_sppf_pool_is_exact(pkg, eng64):
    """max is exact in any precision: y1,y2,y3 must equal the 5/9/13 clipped-window maxima of the fp16 x bit for bit."""
    cat = eng64.read_buffer("backbone.sppf.cat")          # [4*hid, h, w] = [x | y1 | y2 | y3]
    hid = cat.shape[0] // 4
    x = cat[:hid]
    def pool(t, r):
        out = np.empty_like(t)
        h, w = t.shape[1:]
        for y in range(h):
            for xx in range(w):
                out[:, y, xx] = t[:, max(0, y - r):y + r + 1, max(0, xx - r):xx + r + 1].max(axis=(1, 2))
        return out
    for i, r in enumerate((2, 4, 6), start=1):
        assert np.array_equal(cat[i * hid:(i + 1) * hid], pool(x, r)), f"y{i}"


def test_640_heads_vs_reference_fixture(pkg, eng640, torch_cuda):
    gold = load_golden("frame640_seed1234.npz")
    heads = eng640.forward(_frame(pkg, torch_cuda, 1234, 640))
    for name in pkg.graph.OUTPUT_NAMES:
        ref = gold[f"head/{name}"]
        err = heads[name] - ref
        assert np.abs(err).max() < HEAD_ATOL, (name, float(np.abs(err).max()))
        assert np.sqrt((err ** 2).mean()) < (CLS_RMS if name.endswith("cls") else REG_RMS), name


@pytest.mark.parametrize("conf,iou,q", [(0.5, 0.45, 0.1), (0.5, 0.45, 0.0), (0.3, 0.2, 0.0), (0.2, 0.6, 0.25),
                                        (0.0, 0.45, 0.1), (0.9999999, 0.45, 0.1)])
def test_postprocess_exact_on_reference_heads(pkg, eng640, oracle_mod, torch_cuda, conf, iou, q):
    """Identical inputs (the reference's own head tensors) -> the fused kernel must reproduce the oracle
    (engine semantics) record for record: same count, same order, bit-identical boxes and classes."""
    gold = load_golden("frame640_seed1234.npz")
    heads = [gold[f"head/{n}"] for n in pkg.graph.OUTPUT_NAMES]
    for n, h in zip(pkg.graph.OUTPUT_NAMES, heads):
        eng640.outputs[n].copy_(torch_cuda.from_numpy(h)[None])
    got = eng640.postprocess(conf, iou, q)
    want, ncand = oracle_mod.postprocess(heads, conf, iou, q)
    assert len(got) == len(want), (len(got), len(want), ncand)
    if len(want) == 0:
        return
    # GPU expf vs glibc expf may differ by an ulp: confidences within 2e-7; order ties can then swap neighbours,
    # so compare as sets keyed by the (exact) box
    np.testing.assert_allclose(np.sort(got["confidence"]), np.sort(want["confidence"]), atol=2e-7, rtol=0)
    ka = np.lexsort((got["y2"], got["x2"], got["y1"], got["x1"], got["class_id"]))
    kb = np.lexsort((want["y2"], want["x2"], want["y1"], want["x1"], want["class_id"]))
    for f in ("x1", "y1", "x2", "y2", "class_id"):
        assert np.array_equal(got[f][ka], want[f][kb]), f
    assert np.all(np.diff(got["confidence"]) <= 0)
    assert np.all(got["valid"] == 1) and np.all(got["_pad"] == 0)


@pytest.mark.parametrize("case", ["all_equal", "untrained", "saturated", "two_values"])
def test_postprocess_overflow_selection_on_degenerate_heads(pkg, eng640, oracle_mod, torch_cuda, case):
    """More than MAX_DETECTIONS candidates whose confidences crowd into ONE histogram bin -- what an untrained network
    produces (every logit ~ 0, all 33 600 cells pass 0.5): the selection must split the bin again
    (by key, and once the keys are equal by enumeration index) and still return exactly the oracle's 1024 best, ties by
    enumeration order."""
    rng = np.random.default_rng(11)
    heads = []
    for s in (160, 80, 40):
        if case == "all_equal":           # one confidence for every cell: the cut is decided by the enumeration index alone
            cls = np.zeros((4, s, s), np.float32)
            cls[1:] = -1.0
        elif case == "untrained":         # logits ~ 0 +- 1e-3: half the frame inside one 1/4096-wide confidence bin
            cls = rng.normal(0.0, 1e-3, (4, s, s)).astype(np.float32)
        elif case == "saturated":         # confidences pile up at 1.0 (the last bin, its upper end)
            cls = np.where(rng.random((4, s, s)) < 0.6, 30.0, rng.normal(12.0, 1.0, (4, s, s))).astype(np.float32)
        else:                             # two distinct confidences, thousands of cells each
            cls = np.full((4, s, s), -2.0, np.float32)
            cls[0] = np.where(rng.random((s, s)) < 0.5, 0.25, 0.125).astype(np.float32)
        # small boxes on a lattice: nothing overlaps, so the NMS keeps all 1024 and the comparison sees the selection itself
        reg = np.full((4, s, s), 0.2, np.float32)
        heads += [cls, reg]
    for n, h in zip(pkg.graph.OUTPUT_NAMES, heads):
        eng640.outputs[n].copy_(torch_cuda.from_numpy(h)[None])
    thr = 0.5 if case != "two_values" else 0.52
    got = eng640.postprocess(thr, 0.45, 0.0)
    want, ncand = oracle_mod.postprocess(heads, thr, 0.45, 0.0)
    assert ncand > 4000 and len(want) == 1024 and len(got) == 1024, (ncand, len(want), len(got))
    np.testing.assert_allclose(np.sort(got["confidence"]), np.sort(want["confidence"]), atol=2e-7, rtol=0)
    ka = np.lexsort((got["y2"], got["x2"], got["y1"], got["x1"], got["class_id"]))
    kb = np.lexsort((want["y2"], want["x2"], want["y1"], want["x1"], want["class_id"]))
    if case in ("all_equal", "two_values"):   # exact ties: the SAME cells must be chosen (enumeration order), in the same output order
        for f in ("x1", "y1", "x2", "y2", "class_id"):
            assert np.array_equal(got[f], want[f]), f
    else:                                 # (an ulp of expf may move a cell across the cut: compare all but a handful)
        a = set(zip(got["x1"][ka].tolist(), got["y1"][ka].tolist(), got["class_id"][ka].tolist()))
        b = set(zip(want["x1"][kb].tolist(), want["y1"][kb].tolist(), want["class_id"][kb].tolist()))
        assert len(a ^ b) <= 16, len(a ^ b)
    assert np.all(np.diff(got["confidence"]) <= 0) and np.all(got["_pad"] == 0) and np.all(got["valid"] == 1)


def test_postprocess_heavy_overlap_nms(pkg, eng640, oracle_mod, torch_cuda):
    """Synthetic heads where most candidates overlap (big boxes, two classes): exercises the suppression masks."""
    rng = np.random.default_rng(5)
    heads = []
    for s in (160, 80, 40):
        cls = rng.normal(-4.0, 1.0, (4, s, s)).astype(np.float32)
        hot = rng.random((s, s)) < (300.0 / (3 * s * s))
        cls[rng.integers(0, 2, (s, s)), np.arange(s)[:, None], np.arange(s)[None, :]] += np.where(hot, 7.0, 0.0).astype(np.float32)
        reg = rng.uniform(4.0, 9.0, (4, s, s)).astype(np.float32)
        heads += [cls, reg]
    for n, h in zip(pkg.graph.OUTPUT_NAMES, heads):
        eng640.outputs[n].copy_(torch_cuda.from_numpy(h)[None])
    got = eng640.postprocess(0.5, 0.45, 0.1)
    want, ncand = oracle_mod.postprocess(heads, 0.5, 0.45, 0.1)
    assert ncand > 150 and len(want) < ncand * 0.8           # the NMS really suppresses
    assert len(got) == len(want)
    ka = np.lexsort((got["y2"], got["x2"], got["y1"], got["x1"], got["class_id"]))
    kb = np.lexsort((want["y2"], want["x2"], want["y1"], want["x1"], want["class_id"]))
    for f in ("x1", "y1", "x2", "y2", "class_id"):
        assert np.array_equal(got[f][ka], want[f][kb]), f


@pytest.mark.parametrize("q", [0.1, 0.0])
def test_infer_end_to_end_vs_oracle(pkg, eng640, oracle_mod, oracle_sd7, torch_cuda, q):
    """The north-star check: fused forward+decode+NMS vs the fp32 oracle on the same frame."""
    for seed in (1234, 1235):
        x = pkg.rng.frame(seed, 640, 640)
        got = eng640.infer(torch_cuda.from_numpy(x).cuda(), 0.5, 0.45, q)
        o = oracle_mod.forward(oracle_sd7, x)
        want, _ = oracle_mod.postprocess([o[n] for n in pkg.graph.OUTPUT_NAMES], 0.5, 0.45, q)
        check_fp16_detections(got, want, 0.5)


def test_fp16_engine_tail_is_the_format_floor(pkg, sd7, eng640, oracle_mod, oracle_sd7, torch_cuda):
    """The HIP engine's worst detection vs the worst detection of the bit-level fp16 emulation of the same op table
    (torch CPU, tests/emulate.py), both against the fp32 oracle on the same four frames: the engine may not be further
    from fp32 than the arithmetic it implements is. Together with tools/fp16_error_budget.py (which shows that no
    subset of layers short of all of them removes the tail) this bounds the fp16 engine's miss of the north-star
    tolerance formally: it IS the format's floor."""
    from emulate import run_op_table
    from unina_yolo_dla_amd import export
    b = export.EngineBuilder(sd7)
    worst = {"gpu": [0.0, 1.0], "emu": [0.0, 1.0]}
    for seed in (1234, 1235, 1236, 1237):
        x = pkg.rng.frame(seed, 640, 640)
        o = oracle_mod.forward(oracle_sd7, x)
        want, _ = oracle_mod.postprocess([o[n] for n in pkg.graph.OUTPUT_NAMES], 0.5, 0.45, 0.1)
        emu, _ = run_op_table(b, x, fp16=True)
        e_dets, _ = oracle_mod.postprocess([emu[n] for n in pkg.graph.OUTPUT_NAMES], 0.5, 0.45, 0.1)
        g_dets = eng640.infer(torch_cuda.from_numpy(x).cuda(), 0.5, 0.45, 0.1)
        for key, dets in (("gpu", g_dets), ("emu", e_dets)):
            st = compare(dets, want, 0.5, min_iou=0.99, score_tol=1e-2)
            worst[key][0] = max(worst[key][0], st["max_dscore"])
            worst[key][1] = min(worst[key][1], st["min_iou"])
    print("fp16 floor: engine max|ds| %.2e min IoU %.5f | emulation max|ds| %.2e min IoU %.5f" %
          (worst["gpu"][0], worst["gpu"][1], worst["emu"][0], worst["emu"][1]))
    assert worst["gpu"][0] <= 1.3 * worst["emu"][0] + 1e-4, worst
    assert 1.0 - worst["gpu"][1] <= 1.3 * (1.0 - worst["emu"][1]) + 1e-4, worst
    assert worst["emu"][0] > 1e-3                            # the emulation itself misses 1e-3: the floor is the format's


def test_narrow_model_base_channels_16(pkg, oracle_mod, torch_cuda):
    """model.py:331-333 base_channels=16: embedded at width 32 by the exporter (zero channels, exact); heads and
    detections against the fp32 oracle of the NARROW model, same tolerance as the default width."""
    from unina_yolo_dla_amd.engine import Engine
    g = pkg.graph.Graph(base_channels=16, in_h=320, in_w=320)
    sd = pkg.synth.make_state_dict(7, g, head_scales={n: (8.0 if n.endswith("cls") else 2.0) for n in pkg.graph.OUTPUT_NAMES})
    osd = oracle_mod.StateDict(sd)
    e = Engine.from_state_dict(sd, g)
    try:
        assert e.L.unina_fusion_groups(e.h) == 9               # the widened model is graph (A) at width 32
        x = pkg.rng.frame(1234, 320, 320)
        heads = e.forward(torch_cuda.from_numpy(x).cuda())
        o = oracle_mod.forward(osd, x, base_channels=16)
        for n in pkg.graph.OUTPUT_NAMES:
            err = float(np.sqrt(((heads[n] - o[n]) ** 2).mean()))
            assert err < 4e-3 * max(1.0, float(np.abs(o[n]).max())), (n, err)
        thr = 0.2
        got = e.infer(torch_cuda.from_numpy(x).cuda(), thr, 0.45, 0.1)
        want, ncand = oracle_mod.postprocess([o[n] for n in pkg.graph.OUTPUT_NAMES], thr, 0.45, 0.1)
        assert 0 < ncand < 1024
        check_fp16_detections(got, want, thr)
    finally:
        e.close()
        osd.close()


def test_non_square_input_and_empty_result(pkg, sd7, oracle_mod, oracle_sd7, torch_cuda):
    """192 x 320 input (TensorRTEngine::getInputDimensions reports w and h separately, perception_node.cpp:297-325):
    48x80 / 24x40 / 12x20 maps, partial tiles in both directions. Heads vs the fp32 oracle, detections within the fp16
    tolerance, fused = per-op bit for bit, and an empty result (threshold no cell passes) is count 0, not an error."""
    from unina_yolo_dla_amd.engine import Engine
    g = pkg.graph.Graph(in_h=192, in_w=320)
    e = Engine.from_state_dict(sd7, g)
    try:
        x = pkg.rng.frame(1234, 192, 320)
        xd = torch_cuda.from_numpy(x).cuda()
        heads = {k: v.copy() for k, v in e.forward(xd).items()}
        o = oracle_mod.forward(oracle_sd7, x)
        for n in pkg.graph.OUTPUT_NAMES:
            assert heads[n].shape == o[n].shape
            np.testing.assert_allclose(heads[n], o[n], atol=HEAD_ATOL, rtol=0, err_msg=n)
        got = e.infer(xd, 0.5, 0.45, 0.1)
        want, ncand = oracle_mod.postprocess([o[n] for n in pkg.graph.OUTPUT_NAMES], 0.5, 0.45, 0.1)
        assert ncand > 10
        check_fp16_detections(got, want, 0.5)
        assert len(e.infer(xd, 1.0, 0.45, 0.1)) == 0
        e.set_fusion(False)
        plain = e.forward(xd)
        for k in plain:
            same_head(heads[k], plain[k], k)
        from detcmp import compare
        compare(e.infer(xd, 0.5, 0.45, 0.1), got, 0.5, min_iou=0.999, score_tol=1e-3, max_unmatched_frac=0.005)   # (P3 / P4 heads: same_head)
    finally:
        e.close()


@pytest.mark.parametrize("nc,cls_scale,thr", [(1, 30.0, 0.05), (7, 10.0, 0.2)])
def test_other_class_counts(pkg, oracle_mod, torch_cuda, nc, cls_scale, thr):
    """num_classes is a model parameter (model.py:331-333; the node hard-codes 4, perception_node.cpp:630-639): the engine
    takes it from the engine file. 1 and 7 classes (the head's output rows are padded to 16 per branch): heads and
    detections vs the fp32 oracle, fused = per-op (same_head)."""
    from unina_yolo_dla_amd.engine import Engine
    g = pkg.graph.Graph(num_classes=nc, in_h=256, in_w=256)
    sd = pkg.synth.make_state_dict(7, g, head_scales={n: (cls_scale if n.endswith("cls") else 2.0) for n in pkg.graph.OUTPUT_NAMES})
    osd = oracle_mod.StateDict(sd)
    e = Engine.from_state_dict(sd, g)
    try:
        x = pkg.rng.frame(1234, 256, 256)
        xd = torch_cuda.from_numpy(x).cuda()
        heads = {k: v.copy() for k, v in e.forward(xd).items()}
        o = oracle_mod.forward(osd, x, num_classes=nc)
        for n in pkg.graph.OUTPUT_NAMES:
            assert heads[n].shape == o[n].shape and heads[n].shape[0] == (nc if n.endswith("cls") else 4)
            np.testing.assert_allclose(heads[n], o[n], atol=HEAD_ATOL, rtol=0, err_msg=n)
        got = e.infer(xd, thr, 0.45, 0.1)
        want, ncand = oracle_mod.postprocess([o[n] for n in pkg.graph.OUTPUT_NAMES], thr, 0.45, 0.1)
        assert 0 < ncand < 1024 and set(np.unique(got["class_id"])) <= set(range(nc))
        check_fp16_detections(got, want, thr)
        e.set_fusion(False)
        plain = e.forward(xd)
        for k in plain:
            same_head(heads[k], plain[k], k)
    finally:
        e.close()
        osd.close()


def test_infer_matches_reference_fixture_detections(pkg, eng640, torch_cuda):
    """Against detections produced by the reference's own model.py + postprocess.hpp (committed fixture)."""
    gold = load_golden("frame640_seed1234.npz")
    x = _frame(pkg, torch_cuda, 1234, 640)
    for q in (0.1, 0.0):
        got = eng640.infer(x, 0.5, 0.45, q)
        ref = gold[f"ref_dets_q{q}"]
        want = np.zeros(len(ref), dtype=got.dtype)
        for f in ref.dtype.names:
            want[f] = ref[f]
        check_fp16_detections(got, want, 0.5)


def test_fp16_engine_matches_fp16_emulator(pkg, sd7, eng640, torch_cuda):
    """Same arithmetic, different machine: the op table run by torch CPU with fp16 rounding at every buffer write
    (tests/emulate.py) vs the HIP kernels. The two differ only in fp32 summation order (which flips some fp16
    roundings), so: (1) the first layers agree almost bit for bit; (2) the HIP engine is exactly as far from the
    fp32 reference as a plain fp16 PyTorch emulation of the same graph is -- the drift is the format's, not a defect."""
    from emulate import run_op_table
    from unina_yolo_dla_amd import export
    gold = load_golden("frame640_seed1234.npz")
    x = pkg.rng.frame(1234, 640, 640)
    heads = eng640.forward(torch_cuda.from_numpy(x).cuda())
    emu, named = run_op_table(export.EngineBuilder(sd7), x, fp16=True)
    rms = lambda a: float(np.sqrt((a.astype(np.float64) ** 2).mean()))
    for name in pkg.graph.OUTPUT_NAMES:
        ref = gold[f"head/{name}"]
        e_gpu, e_emu, e_x = rms(heads[name] - ref), rms(emu[name] - ref), rms(heads[name] - emu[name])
        assert e_gpu < 1.2 * e_emu + 1e-5, (name, e_gpu, e_emu)      # no worse than the fp16 emulation
        assert e_x < 1.2 * max(e_gpu, e_emu), (name, e_x, e_gpu, e_emu)
    eng640.set_fusion(False)      # (stage1_conv's output stays in LDS when the block kernel computes it: per-op forward)
    try:
        eng640.forward(torch_cuda.from_numpy(x).cuda())
        for bname, lim in (("backbone.stem", 0.01), ("backbone.stage1_conv", 0.03)):
            got = eng640.read_buffer(bname)
            mism = float((got != named[bname]).mean())
            assert mism < lim, (bname, mism)                       # fp16 buffers: identical bits except rare 1-ulp flips
            assert np.abs(got - named[bname]).max() <= 2e-3 * max(1.0, np.abs(named[bname]).max())
    finally:
        eng640.set_fusion(True)


# ---- fp32 precision mode: native fp32 MFMA, same graph, same kernels -> the north-star tolerance holds outright ----
@pytest.fixture(scope="module")
def eng640_fp32(pkg, sd7, torch_cuda):
    from unina_yolo_dla_amd import export
    from unina_yolo_dla_amd.engine import Engine
    e = Engine.from_state_dict(sd7, precision=export.FP32)
    yield e
    e.close()


def test_fp32_engine_heads_match_reference_fixture(pkg, eng640_fp32, torch_cuda):
    gold = load_golden("frame640_seed1234.npz")
    heads = eng640_fp32.forward(_frame(pkg, torch_cuda, 1234, 640))
    for name in pkg.graph.OUTPUT_NAMES:
        np.testing.assert_allclose(heads[name], gold[f"head/{name}"], atol=2e-4, rtol=0, err_msg=name)   # measured ~3e-5


@pytest.mark.parametrize("q", [0.1, 0.0])
def test_fp32_engine_meets_north_star_tolerance(pkg, eng640_fp32, oracle_mod, oracle_sd7, torch_cuda, q):
    """BASELINE.json tolerance, every detection: IoU >= 0.999 and |score delta| < 1e-3 against the fp32 oracle, and
    against the detections of the reference's own model.py + postprocess.hpp (fixture)."""
    gold = load_golden("frame640_seed1234.npz")
    for seed in (1234, 1235):
        x = pkg.rng.frame(seed, 640, 640)
        got = eng640_fp32.infer(torch_cuda.from_numpy(x).cuda(), 0.5, 0.45, q)
        o = oracle_mod.forward(oracle_sd7, x)
        want, _ = oracle_mod.postprocess([o[n] for n in pkg.graph.OUTPUT_NAMES], 0.5, 0.45, q)
        stats = compare(got, want, 0.5, min_iou=0.999, score_tol=1e-3, max_unmatched_frac=0.005)
        assert stats["matched"] >= len(want) - 2 and stats["max_dscore"] < 1e-4, stats
        if seed == 1234:
            ref = gold[f"ref_dets_q{q}"]
            want = np.zeros(len(ref), dtype=got.dtype)
            for f in ref.dtype.names:
                want[f] = ref[f]
            stats = compare(got, want, 0.5, min_iou=0.999, score_tol=1e-3, max_unmatched_frac=0.005)
            assert stats["matched"] >= len(want) - 2, stats


def test_fp32_engine_mini64_buffers(pkg, sd7, oracle_mod, oracle_sd7, torch_cuda):
    from unina_yolo_dla_amd import export
    from unina_yolo_dla_amd.engine import Engine
    e = Engine.from_state_dict(sd7, pkg.graph.Graph(in_h=64, in_w=64), precision=export.FP32)
    try:
        x = pkg.rng.frame(1234, 64, 64)
        heads = e.forward(torch_cuda.from_numpy(x).cuda())
        ref = oracle_mod.forward(oracle_sd7, x, keep_all=True)
        for name in pkg.graph.OUTPUT_NAMES:
            np.testing.assert_allclose(heads[name], ref[name], atol=1e-4, rtol=0, err_msg=name)
        for bname, oname in {"backbone.stem": "backbone.stem", "backbone.sppf.cat": "backbone.sppf.cat",
                             "neck.cat_fpn2": "neck.cat_fpn2", "neck.cat_pan2": "neck.cat_pan2",
                             "p4_out": "neck.pan_c3k2_2.cv3"}.items():
            np.testing.assert_allclose(e.read_buffer(bname), ref[oname], atol=1e-4, rtol=1e-4, err_msg=bname)
    finally:
        e.close()


# ---- INT8 engine (BASELINE config 3): per-tensor symmetric scales, own calibrator, reference carve-outs ----
def test_int8_engine_matches_integer_emulation_and_reports_drift(pkg, sd7, oracle_mod, oracle_sd7, torch_cuda):
    """(1) exactness: the HIP int8 path (v_mfma_i32_16x16x64_i8, fp32 per-channel multiplier, round-half-even
    requantisation) against the torch-CPU integer emulation of the same op table at 128x128, op by op on the engine's
    own inputs (teacher-forced) -- int8 codes must agree but for rare round-to-nearest ties; (2) calibrated drift vs the fp32 oracle at
    640x640 with the histogram + mse calibrator (the reference pins no quantised result and pytorch-quantization is absent:
    PARITY UNPINNED; the per-calibrator drift table is profiles/r02/int8_drift_table.txt)."""
    from emulate import run_op_table, engine_buffers, per_op_mismatch
    from unina_yolo_dla_amd import export
    from unina_yolo_dla_amd.engine import Engine, calibrate_amax
    g = pkg.graph.Graph(in_h=128, in_w=128)
    frames = [pkg.rng.frame(5000 + i, 128, 128) for i in range(4)]
    amax = calibrate_amax(sd7, g, frames)
    b8 = export.EngineBuilder(sd7, g, export.INT8, amax)
    e = Engine.from_state_dict(sd7, g, precision=export.INT8, amax=amax)
    try:
        x = pkg.rng.frame(1234, 128, 128)
        heads = e.forward(torch_cuda.from_numpy(x).cuda())
        emu, named = run_op_table(b8, x)
        # per-op exactness, teacher-forced: every op of the emulation reads the ENGINE's own (per-op forward) buffers, so
        # a rounding flip cannot snowball (free-running, a handful of +-1 input codes moves a third of a deep layer's
        # outputs by one code). int8 codes must agree but for the rare round-to-nearest tie; fp16 values within 2 ulp.
        e.set_fusion(False)
        e.forward(torch_cuda.from_numpy(x).cuda())
        teacher = engine_buffers(b8, e.read_buffer)
        e.set_fusion(True)
        forced = run_op_table(b8, x, teacher=teacher)[1]
        mm = per_op_mismatch(b8, teacher, forced)
        assert len(mm) > 40
        for bname, (frac, worst) in mm.items():
            i8 = b8.buffers[[bb[0] for bb in b8.buffers].index(bname)][4] == export.BUF_I8
            assert frac < 2e-3 and worst <= (1.0 if i8 else 4.0), (bname, i8, frac, worst)
        ref = oracle_mod.forward(oracle_sd7, x)
        for n in pkg.graph.OUTPUT_NAMES:
            # free-running, the engine and the emulation are two equally valid roundings of the same arithmetic (they
            # differ in fp16 summation order only): each must sit at the same distance -- the quantisation drift --
            # from the fp32 oracle, i.e. the kernels add no error of their own
            e_engine = float(np.sqrt(((heads[n] - ref[n]) ** 2).mean()))
            e_quant = float(np.sqrt(((emu[n] - ref[n]) ** 2).mean()))
            assert e_engine < 1.25 * e_quant + 1e-3, (n, e_engine, e_quant)
    finally:
        e.close()
    # ---- drift at the benchmark size (PARITY UNPINNED: the reference pins no quantised result and its quantisation
    # library is absent; bounds = the measured drift of the chosen calibrator, profiles/r03/int8_drift_table.txt, 8 frames:
    # histogram + mse, 97.6 % matched, IoU median 0.9934 / p5 0.9871, |dscore| median 0.0067 / p95 0.022 / max 0.10, 5.5 % of
    # the engine's detections without a partner, head rms 3.4 % of std -- plus a margin). The table also says what the tail IS:
    # every matched pair below IoU 0.9 (11 of 3 677; the "IoU min 0.51") and 198 of the 208 partnerless detections carry a box
    # that equals (IoU >= 0.97) some PRE-NMS oracle candidate of the class -- a keep / suppress decision inside a cluster of
    # overlapping cells fell the other way because two scores swapped order; the geometry itself does not drift. Asserted below. ----
    frames = [pkg.rng.frame(5000 + i, 640, 640) for i in range(8)]
    amax = calibrate_amax(sd7, None, frames, method="mse")
    e = Engine.from_state_dict(sd7, precision=export.INT8, amax=amax)
    try:
        from detcmp import iou_matrix
        ious, dss, n_got, n_want, n_matched, n_low, n_low_flip, n_extra, n_extra_flip = [], [], 0, 0, 0, 0, 0, 0, 0
        for seed in (1234, 1235, 1236):
            x = pkg.rng.frame(seed, 640, 640)
            heads = {k: v.copy() for k, v in e.forward(torch_cuda.from_numpy(x).cuda()).items()}   # (unina_infer computes the P3 / P4
            got = e.infer(torch_cuda.from_numpy(x).cuda(), 0.5, 0.45, 0.1)                          # output convs inside the decode launch)
            o = oracle_mod.forward(oracle_sd7, x)
            want, _ = oracle_mod.postprocess([o[n] for n in pkg.graph.OUTPUT_NAMES], 0.5, 0.45, 0.1)
            cand, _ = oracle_mod.postprocess([o[n] for n in pkg.graph.OUTPUT_NAMES], 0.45, 1.0, 0.1)    # every candidate, no suppression
            for n in pkg.graph.OUTPUT_NAMES:
                err = float(np.sqrt(((heads[n] - o[n]) ** 2).mean()))
                assert err < 0.08 * max(float(o[n].std()), 0.3), (n, err)
            m = iou_matrix(got, want)
            m = np.where(got["class_id"][:, None] == want["class_id"][None, :], m, 0.0)
            j = m.argmax(1)
            matched = m.max(1) > 0.5
            mc = iou_matrix(got, cand)
            mc = np.where(got["class_id"][:, None] == cand["class_id"][None, :], mc, 0.0).max(1)
            low = matched & (m.max(1) < 0.9)
            ious += list(m.max(1)[matched]); dss += list(np.abs(got["confidence"] - want["confidence"][j])[matched])
            n_got += len(got); n_want += len(want); n_matched += int(matched.sum())
            n_low += int(low.sum()); n_low_flip += int((low & (mc >= 0.97)).sum())
            n_extra += int((~matched).sum()); n_extra_flip += int(((~matched) & (mc >= 0.97)).sum())
        ious, dss = np.array(ious), np.array(dss)
        print("INT8 drift: dets", n_got, "vs", n_want, "matched", n_matched, "IoU median %.4f p5 %.4f min %.4f" % (np.median(ious), np.percentile(ious, 5), ious.min()),
              "|dscore| median %.4f p95 %.4f max %.4f" % (np.median(dss), np.percentile(dss, 95), dss.max()),
              "IoU<0.9:", n_low, "(flips:", n_low_flip, ") partnerless:", n_extra, "(flips:", n_extra_flip, ")")
        assert n_matched >= 0.94 * n_want and np.median(ious) > 0.985
        assert np.median(dss) < 0.012
        # the tail, bounded: measured p5 0.9871, |dscore| p95 0.022 / max 0.10, partnerless 5.5 % of the oracle's count
        assert np.percentile(ious, 5) > 0.975 and np.percentile(dss, 95) < 0.04 and dss.max() < 0.2
        assert n_extra <= 0.10 * n_want and abs(n_got - n_want) <= 0.08 * n_want
        # ... and explained: a poorly matched box is a cluster decision, not geometry (every one of them; >= 85 % of the partnerless)
        assert n_low <= 0.01 * n_matched and n_low_flip == n_low, (n_low, n_low_flip)
        assert n_extra_flip >= 0.85 * n_extra, (n_extra, n_extra_flip)
    finally:
        e.close()


def test_determinism_and_rebinding(pkg, eng640, torch_cuda):
    xs = [_frame(pkg, torch_cuda, s, 640) for s in (1, 2)]
    first = [eng640.infer(x).tobytes() for x in xs]
    for _ in range(5):
        for x, f in zip(xs, first):
            assert eng640.infer(x).tobytes() == f            # bit-identical across replays and input re-binding
    assert first[0] != first[1]


@pytest.mark.parametrize("precision", ["fp16", "int8"])
def test_tile_configs_and_autotune_are_bit_identical(pkg, sd7, torch_cuda, precision):
    """Every conv tile configuration accumulates each output's K terms in the same order (fp16) or exactly (int8), so
    forcing any fitting configuration -- im2col, halo and register-queue kernels, stride 1 and 2 -- or letting the
    autotuner choose must not change a single bit of the outputs. The one exception is the fp16 chunked weights-stationary
    kernel (conv3x3_ws<f16,...>, conv_igemm.hip conv3x3_wsc_body): it sums chunk-major, its fp32 accumulators differ in the
    last bits and an fp16 output flips its last place now and then -- wherever it runs the heads must agree within 5e-3
    (the fp16 format's own noise is 3e-3 on a score)."""
    from unina_yolo_dla_amd import export
    from unina_yolo_dla_amd.engine import Engine, calibrate_amax
    g = pkg.graph.Graph(in_h=128, in_w=128)
    if precision == "int8":
        amax = calibrate_amax(sd7, g, [pkg.rng.frame(5000 + i, 128, 128) for i in range(2)])
        e = Engine.from_state_dict(sd7, g, precision=export.INT8, amax=amax)
    else:
        e = Engine.from_state_dict(sd7, g)
    try:
        x = _frame(pkg, torch_cuda, 1234, 128)
        fused = {k: v.copy() for k, v in e.forward(x).items()}
        e.set_fusion(False)                                   # tile configurations belong to the per-op kernels
        base = {k: v.copy() for k, v in e.forward(x).items()}
        for k in base:
            if precision == "fp16":   # (the fused frame runs the P3 | P4 head pairs on the chunked kernel)
                np.testing.assert_allclose(fused[k], base[k], atol=5e-3, rtol=0, err_msg=f"fusion {k}")
            else:
                assert np.array_equal(fused[k], base[k]), ("fusion", k)
        infos = e.op_infos()
        ncfg = len(e.conv_configs())
        tried = 0
        for cfg in range(ncfg):
            applied = [i for i, o in enumerate(infos) if o["kind"] == 1 and e.set_op_config(i, cfg)]
            if not applied:
                continue
            tried += 1
            out = e.forward(x)
            chunked = "conv3x3_ws<f16" in e.conv_configs()[cfg]
            for k in base:
                if chunked:
                    np.testing.assert_allclose(out[k], base[k], atol=5e-3, rtol=0, err_msg=f"{e.conv_configs()[cfg]} {k}")
                else:
                    assert np.array_equal(out[k], base[k]), (e.conv_configs()[cfg], k)
            for i in applied:
                e.set_op_config(i, -1)
        assert tried >= 6
        e.autotune(x, iters=3)
        out = e.forward(x)
        chunked = any("conv3x3_ws<f16" in o["kernel"] for o in e.op_infos())
        for k in base:
            if chunked:
                np.testing.assert_allclose(out[k], base[k], atol=5e-3, rtol=0, err_msg=f"autotune {k}")
            else:
                assert np.array_equal(out[k], base[k]), ("autotune", k)
    finally:
        e.close()


# With fusion on, a down-sampling conv (down1 / down2) runs as the first step of the PAN block that consumes it and its
# output -- the first channels of the concat buffer -- never reaches HBM: those channels are compared through the block's
# output only.
WRITTEN_WHEN_FUSED = {"neck.cat_pan1": (64, None), "neck.cat_pan2": (128, None), "cat_pan1": (64, None), "cat_pan2": (128, None),
                      "backbone.sppf.cat": (0, 128)}    # (the three pooled maps of the SPPF are formed in LDS by the cv2 / lateral kernel)


def written(bname, arr):
    a, b = WRITTEN_WHEN_FUSED.get(bname, (0, None))
    return arr[a:b]


BLOCK_OUTPUTS = ("neck.cat_fpn2", "neck.cat_fpn1", "neck.cat_pan2", "neck.cat_pan1", "p2_fused", "p3_out", "p4_out",
                 "backbone.sppf.cat")    # (sppf.cv1 runs as the last step of stage3's block kernel)


def same_head(a, b, name):
    """fp16 engines, fused frame vs per-op table: the P2 head must agree bit for bit; the P3 / P4 heads run their two 3x3
    layers on the chunked weights-stationary pair kernel in the fused frame (chunk-major sum order, conv_igemm.hip
    conv3x3_wsc_body: last-bit differences of fp16 outputs) and agree within 5e-3."""
    if name.startswith("p2_"):
        assert np.array_equal(a, b), name
    else:
        np.testing.assert_allclose(a, b, atol=5e-3, rtol=0, err_msg=name)


@pytest.mark.parametrize("size", [64, 640, 96])
def test_block_fusion_is_bit_identical(pkg, sd7, torch_cuda, size):
    """Each C3k2 block (model.py:76-110) and the P2 DetectionHead (model.py:274-303) as ONE launch with its
    intermediates in LDS (csrc/c3k2_fused.hip, head_fused.hip) vs the per-conv launches: same MFMA, same K order, same fp16 rounding points -> every block output and the P2
    head must agree bit for bit (the P3 / P4 heads: same_head). 96x96 gives 24/12/6-pixel maps: partial tiles on every level."""
    from unina_yolo_dla_amd.engine import Engine
    e = Engine.from_state_dict(sd7, pkg.graph.Graph(in_h=size, in_w=size))
    try:
        assert e.L.unina_fusion_groups(e.h) == 9              # 7 C3k2 blocks + the P2 head + the sppf.cv2 -> lateral_p3 pair, on by default
        x = _frame(pkg, torch_cuda, 1234, size)
        fused = {k: v.copy() for k, v in e.forward(x).items()}
        fused_bufs = {b: e.read_buffer(b) for b in BLOCK_OUTPUTS}
        kernels = [o["kernel"] for o in e.op_infos()]
        assert sum("c3k2_fused" in k or "block_dual" in k for k in kernels) == 7    # (one block may share its grid with the head)
        assert sum("head_fused" in k or "block_dual" in k for k in kernels) == 1
        assert sum(k.startswith("conv_dual") for k in kernels) == 3                           # P3 | P4 head layers pairwise
        assert sum(k.startswith("conv_pair") for k in kernels) == 1                           # sppf.cv2 -> lateral_p3
        assert e.set_fusion(False) == 0
        plain = e.forward(x)
        for b in BLOCK_OUTPUTS:
            assert np.array_equal(written(b, fused_bufs[b]), written(b, e.read_buffer(b))), b
        for k in plain:
            same_head(fused[k], plain[k], k)
        assert e.set_fusion(True) == 9
        again = e.forward(x)
        for k in plain:
            assert np.array_equal(again[k], fused[k]), k
            same_head(again[k], plain[k], k)
    finally:
        e.close()


@pytest.mark.parametrize("size", [128, 640, 96])
def test_int8_block_fusion_is_bit_identical(pkg, sd7, torch_cuda, size):
    """INT8 engines: the five all-int8 C3k2 blocks as int8 block kernels (csrc/block_kernels.h with EltI8: LDS images
    hold int8 codes, v_mfma_i32_16x16x64_i8, the per-op kernels' fma / ReLU / shortcut / round-half-even epilogue with
    the same per-tensor scales) vs the per-conv launches: int32 accumulation is exact, so every code must agree."""
    from unina_yolo_dla_amd import export
    from unina_yolo_dla_amd.engine import Engine, calibrate_amax
    g = pkg.graph.Graph(in_h=size, in_w=size)
    amax = calibrate_amax(sd7, g, [pkg.rng.frame(5000 + i, size, size) for i in range(2)])
    e = Engine.from_state_dict(sd7, g, precision=export.INT8, amax=amax)
    try:
        assert e.L.unina_fusion_groups(e.h) == 9              # 5 int8 blocks + the 2 narrow fp16 blocks + the fp16 P2 head + the int8 conv pair
        x = _frame(pkg, torch_cuda, 1234, size)
        fused = {k: v.copy() for k, v in e.forward(x).items()}
        bufs = ("neck.cat_fpn1", "neck.cat_pan1", "neck.cat_pan2", "p3_out", "p4_out", "backbone.sppf.cat",
                "neck.cat_fpn2", "p2_fused", "p2_fused.q8")   # (int8 block -> fp16 lateral tail; fp16 block -> int8 twin store)
        names = [b[0] for b in export.EngineBuilder(sd7, g, export.INT8, amax).buffers]
        bufs = tuple(b for b in bufs if b in names)
        assert len(bufs) == 9
        fused_bufs = {b: e.read_buffer(b) for b in bufs}
        kernels = [o["kernel"] for o in e.op_infos()]
        assert sum("c3k2_fused<i8" in k or "c3k2i8" in k for k in kernels) == 5, kernels
        assert not any("quant_f16_i8" in k for k in kernels) and sum("lat f16" in k for k in kernels) == 1, kernels
        assert e.set_fusion(False) == 0
        plain = e.forward(x)
        for b in bufs:
            assert np.array_equal(written(b, fused_bufs[b]), written(b, e.read_buffer(b))), b
        for k in plain:
            assert np.array_equal(fused[k], plain[k]), k
        assert e.set_fusion(True) == 9
        d1 = e.infer(x, 0.5, 0.45, 0.1)
        e.set_fusion(False)
        d0 = e.infer(x, 0.5, 0.45, 0.1)
        assert len(d0) == len(d1) and d0.tobytes() == d1.tobytes()
    finally:
        e.close()


def test_frame_graph_and_tiled_nms_match_the_plain_forms(pkg, sd7, torch_cuda, monkeypatch):
    """Default per-frame path = ONE hipGraph launch (stem + forward + two-launch post-process whose NMS mask tiles run on
    many CUs; the stem / post-process nodes are re-pointed per frame with hipGraphExecKernelNodeSetParams). It must
    return exactly what separate launches with the one-workgroup post-process return -- also when thresholds and input
    buffers change between frames, and when more than MAX_DETECTIONS cells pass (top-1024 selection)."""
    from unina_yolo_dla_amd.engine import Engine
    calls = [(1234, 0.5, 0.45, 0.1), (7, 0.5, 0.45, 0.1), (1234, 0.3, 0.6, 0.0), (9, 0.02, 0.45, 0.1), (1234, 0.5, 0.45, 0.1)]
    frames = {s: _frame(pkg, torch_cuda, s, 640) for s in {c[0] for c in calls}}

    def run():
        e = Engine.from_state_dict(sd7)
        try:
            out = [e.infer(frames[s], c, i, q).tobytes() for s, c, i, q in calls]
            return out, e.L.unina_fusion_groups(e.h)
        finally:
            e.close()

    fast, groups = run()
    assert groups == 9
    monkeypatch.setenv("UNINA_FULL_GRAPH", "0")
    monkeypatch.setenv("UNINA_POST_SPLIT", "0")
    plain, groups = run()
    assert groups == 9
    assert fast == plain
    assert len(fast[3]) > len(fast[0]) and fast[0] == fast[4] and fast[0] != fast[1]
    # ... and the per-op table on top (no block kernels, no pair launches: the P3 / P4 head layers on the im2col kernels, whose
    # sum order differs from the chunked pair kernel's in the last bits -- same_head): every detection inside the north-star tolerance
    monkeypatch.setenv("UNINA_FUSE", "0")
    perop, groups = run()
    assert groups == 0
    from detcmp import compare
    from unina_yolo_dla_amd.engine import DET_DTYPE
    for (s_, c, i, q), a_, b_ in zip(calls, fast, perop):
        if c >= 0.1:   # (the 0.02 call overflows MAX_DETECTIONS: the top-1024 cut is compared exactly above)
            compare(np.frombuffer(a_, DET_DTYPE), np.frombuffer(b_, DET_DTYPE), c, min_iou=0.999, score_tol=1e-3, max_unmatched_frac=0.005, iou_thr=i)


def test_async_result_layout(pkg, eng640, torch_cuda):
    x = _frame(pkg, torch_cuda, 1234, 640)
    sync = eng640.infer(x)
    buf = eng640.infer_async(x)
    torch_cuda.cuda.synchronize()
    assert eng640.unpack(buf).tobytes() == sync.tobytes()


def test_stepwise_reference_api(pkg, eng640, oracle_mod, torch_cuda):
    """perception_node.cpp:627-656 call sequence through the seven gpu_postprocess.h symbols."""
    import ctypes as C
    L = eng640.L
    gold = load_golden("frame640_seed1234.npz")
    heads = [gold[f"head/{n}"] for n in pkg.graph.OUTPUT_NAMES]
    dev = [torch_cuda.from_numpy(h).cuda() for h in heads]
    dets = torch_cuda.zeros(1024 * 8, dtype=torch_cuda.int32, device="cuda")
    assert L.init_postprocess_resources() == 0
    try:
        stream = torch_cuda.cuda.current_stream().cuda_stream
        assert L.reset_detection_counter(stream) == 0
        for i, s in enumerate((4, 8, 16)):
            c, r = dev[2 * i], dev[2 * i + 1]
            assert L.decode_yolo_head(c.data_ptr(), r.data_ptr(), dets.data_ptr(), c.shape[2], c.shape[1], s, 4,
                                      0.5, 0.1, stream) == 0
        n = C.c_int(-1)
        assert L.get_detection_count(C.byref(n), stream) == 0
        torch_cuda.cuda.synchronize()
        want, ncand = oracle_mod.postprocess(heads, 0.5, 0.45, 0.1)
        assert n.value == ncand
        n = min(n.value, 1024)
        assert L.run_gpu_nms(dets.data_ptr(), n, 0.45, stream) == 0
        from unina_yolo_dla_amd.engine import DET_DTYPE
        host = np.zeros(1024, dtype=DET_DTYPE)
        valid = C.c_int(-1)
        assert L.copy_valid_detections_to_host(dets.data_ptr(), host.ctypes.data, n, C.byref(valid), stream) == 0
        got = host[:valid.value]
        assert len(got) == len(want)
        ka = np.lexsort((got["y2"], got["x2"], got["y1"], got["x1"], got["class_id"]))
        kb = np.lexsort((want["y2"], want["x2"], want["y1"], want["x1"], want["class_id"]))
        for f in ("x1", "y1", "x2", "y2", "class_id"):
            assert np.array_equal(got[f][ka], want[f][kb]), f
    finally:
        L.cleanup_postprocess_resources()


def test_1280_p2_head_config(pkg, sd7, torch_cuda):
    """BASELINE config 5: 1280x1280 (P2 grid 320x320). Checked against the reference's sampled head values."""
    from unina_yolo_dla_amd.engine import Engine
    gold = load_golden("frame1280_seed1234.npz")
    e = Engine.from_state_dict(sd7, pkg.graph.Graph(in_h=1280, in_w=1280))
    try:
        x = _frame(pkg, torch_cuda, 1234, 1280)
        heads = e.forward(x)
        for name in pkg.graph.OUTPUT_NAMES:
            flat = heads[name].reshape(-1)
            np.testing.assert_allclose(flat[gold[f"idx/{name}"]], gold[f"vals/{name}"], atol=HEAD_ATOL, rtol=0, err_msg=name)
            assert abs(float(flat.astype(np.float64).mean()) - gold[f"stats/{name}"][0]) < 2e-3
        np.testing.assert_allclose(heads["p4_cls"], gold["head/p4_cls"], atol=HEAD_ATOL, rtol=0)
        got = e.infer(x, 0.6, 0.45, 0.1)
        ref = gold["ref_dets_conf0.6_q0.1"]
        want = np.zeros(len(ref), dtype=got.dtype)
        for f in ref.dtype.names:
            want[f] = ref[f]
        check_fp16_detections(got, want, 0.6)
    finally:
        e.close()
