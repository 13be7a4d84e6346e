"""Graph (B) -- the reference's QAT model UNINA_YOLO_DLA_QAT (unina_yolo_dla/qat.py:350-491) -- on the CPU side:
the oracle's restatement against fixtures made by importing the reference qat.py as-is
(tests/golden/make_golden_qat.py; float fallback, qat.py:249-254), the exporter's op table against that oracle, the
checkpoint key import, and the load-time block matcher."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from emulate import run_op_table

FWD_ATOL = 1e-4


@pytest.fixture(scope="module")
def sd7b(pkg):
    return pkg.synth.make_state_dict(7, pkg.graph.Graph(variant="B"))


@pytest.fixture(scope="module")
def osd7b(oracle_mod, sd7b):
    h = oracle_mod.StateDict(sd7b)
    yield h
    h.close()


def test_graph_b_matches_the_reference_facts(pkg):
    g = pkg.graph.Graph(variant="B")
    assert len(g.convs()) == 74                                # 74 convs, 7 170 872 params
    shapes = g.param_shapes()
    assert sum(int(np.prod(s)) for k, s in shapes.items() if "running" not in k) == 7_170_872
    assert int(load_golden("qat_mini64_seed1234.npz")["n_params"]) == 7_170_872      # counted on the reference model
    assert shapes["stem.conv.weight"] == (32, 3, 3, 3) and shapes["stage4_conv.conv.weight"] == (512, 256, 3, 3)
    assert shapes["head_p4_reg.2.bias"] == (4,) and "head_p2_cls.0.bn.running_var" in shapes
    assert not set(shapes) & set(pkg.graph.Graph().param_shapes())               # zero keys in common with graph (A)
    with pytest.raises(ValueError):
        pkg.graph.Graph(variant="B", in_h=48, in_w=64)                           # stride-32 stage: multiples of 32


def test_oracle_graph_b_mini64_every_module(pkg, oracle_mod, osd7b):
    gold = load_golden("qat_mini64_seed1234.npz")
    x = pkg.rng.frame(1234, 64, 64)
    out = oracle_mod.forward(osd7b, x, keep_all=True, variant="B")
    for name in pkg.graph.OUTPUT_NAMES:
        np.testing.assert_allclose(out[name], gold[f"head/{name}"], atol=FWD_ATOL, rtol=0)
    checked = 0
    for key in gold.files:
        if not key.startswith("tap/"):
            continue
        name = key[4:]
        oname = name + ".add" if name.split(".")[-2:-1] == ["bottlenecks"] else name
        assert oname in out, name
        np.testing.assert_allclose(out[oname], gold[key].astype(np.float32), atol=2e-3, rtol=2e-3, err_msg=name)
        checked += 1
    assert checked >= 80


def test_oracle_graph_b_640_heads_and_reference_detections(pkg, oracle_mod, osd7b):
    gold = load_golden("qat_frame640_seed1234.npz")
    x = pkg.rng.frame(1234, 640, 640)
    out = oracle_mod.forward(osd7b, x, variant="B")
    for name in pkg.graph.OUTPUT_NAMES:
        np.testing.assert_allclose(out[name], gold[f"head/{name}"], atol=FWD_ATOL, rtol=0, err_msg=name)
    thr = float(gold["conf_thr"])
    dets, ncand = oracle_mod.postprocess([gold[f"head/{n}"] for n in pkg.graph.OUTPUT_NAMES], thr, 0.45, 0.1,
                                         sem=oracle_mod.semantics("cpu"))
    ref = gold["ref_dets_q0.1"]
    assert ncand == int(gold["ref_ncand_q0.1"]) and len(dets) == len(ref)       # the reference's postprocess.hpp on the same heads


def test_op_table_graph_b_fp32_equals_oracle(pkg, sd7b, oracle_mod, osd7b):
    from unina_yolo_dla_amd import export
    torch.set_num_threads(4)
    g = pkg.graph.Graph(variant="B", in_h=64, in_w=64)
    b = export.EngineBuilder(sd7b, g)
    assert len(b.ops) == 58                                    # 74 convs -> 56 conv launches + stem + pool
    x = pkg.rng.frame(1234, 64, 64)
    outs, named = run_op_table(b, x, fp16=False)
    ref = oracle_mod.forward(osd7b, x, keep_all=True, variant="B")
    for n in pkg.graph.OUTPUT_NAMES:
        np.testing.assert_allclose(outs[n], ref[n], atol=1.5e-2, rtol=0, err_msg=n)
    for bname, oname in {"stem": "stem", "cat_fpn1": "cat_fpn1", "cat_pan2": "cat_pan2", "stage4_sppf.cat": "stage4_sppf.cat",
                         "stage4_sppf": "stage4_sppf.cv2", "p2_fused": "fpn_c3k2_3.cv3", "p4_out": "pan_c3k2_2.cv3"}.items():
        np.testing.assert_allclose(named[bname], ref[oname], atol=1e-2 * max(1.0, np.abs(ref[oname]).max()), rtol=0, err_msg=bname)


def test_qat_checkpoint_key_import(pkg, sd7b):
    """A pytorch-quantization checkpoint of the QAT model carries quantizer state next to the weights
    (`<conv>._input_quantizer._amax`, `<conv>._weight_quantizer._amax`, `<bottleneck>.residual_quantizer._amax`;
    name rules qat.py:554-563, 657-673) plus BatchNorm `num_batches_tracked`: the importer separates them."""
    from unina_yolo_dla_amd import statedict
    ck = dict(sd7b)
    ck["stage2_conv.conv._input_quantizer._amax"] = np.array(3.5, dtype=np.float32)
    ck["stage2_conv.conv._weight_quantizer._amax"] = np.array(0.25, dtype=np.float32)
    ck["stage2_c3k2.bottlenecks.0.residual_quantizer._amax"] = np.array(6.0, dtype=np.float32)
    ck["stem.bn.num_batches_tracked"] = np.array(12, dtype=np.int64)
    ck = {("module." + k): v for k, v in ck.items()}                           # DataParallel-style prefix
    weights, q = statedict.from_qat_checkpoint(ck)
    assert set(weights) == set(sd7b) and all(np.array_equal(weights[k], sd7b[k]) for k in sd7b)
    assert q["input_amax"] == {"stage2_conv": 3.5} and q["weight_amax"] == {"stage2_conv": 0.25}
    assert q["residual_amax"] == {"stage2_c3k2.bottlenecks.0": 6.0}
    assert statedict.detect_variant(weights) == "B" and statedict.detect_variant(pkg.synth.make_state_dict(7)) == "A"


def test_graph_b_blocks_are_recognised_at_load(pkg, sd7b, tmp_path):
    from unina_yolo_dla_amd import build, engine, export
    build.build_native()
    lib = engine.load_library()
    path = str(tmp_path / "b.une")
    export.export_engine(sd7b, path, pkg.graph.Graph(variant="B", in_h=64, in_w=64))
    assert lib.unina_debug_fusable_groups(path.encode()) == 9   # 8 C3k2 blocks + the P2 head


def _synthetic_qat_checkpoint(pkg, sd7b, g, n_frames=3):
    """What a calibrated pytorch-quantization checkpoint of the QAT model holds, built from float statistics: one
    `_input_quantizer._amax` per QuantConv2d (max |input| over the calibration frames), one `_weight_quantizer._amax`
    (max |W|), next to the weights."""
    from unina_yolo_dla_amd import export
    b16 = export.EngineBuilder(sd7b, g)
    per_buf = export.calibrate(run_op_table(b16, pkg.rng.frame(5000 + i, g.in_h, g.in_w))[1] for i in range(n_frames))
    ck = dict(sd7b)
    for op in b16.ops:
        if op.kind != export.OP_CONV:
            continue
        for sg in op.segs:
            if not sg.bn:
                continue                                   # the heads' output convs are plain nn.Conv2d (qat.py:416): no quantizers
            ck[f"{sg.module}.conv._input_quantizer._amax"] = np.float32(per_buf[b16.buffers[op.src_buf][0]])
            ck[f"{sg.module}.conv._weight_quantizer._amax"] = np.float32(np.abs(sd7b[f"{sg.module}.conv.weight"]).max())
    return ck, per_buf


def test_int8_engine_table_from_a_qat_checkpoint(pkg, sd7b, oracle_mod, osd7b, tmp_path):
    """"INT8 weights from qat.py consumed directly": the checkpoint's own quantizer ranges become the engine's buffer /
    weight scales (one scale per buffer = the largest range among the convs reading it), the reference's FP16
    carve-outs hold for qat.py's module names, and the integer emulation stays within PTQ-typical drift of the oracle."""
    from unina_yolo_dla_amd import export, statedict
    g = pkg.graph.Graph(variant="B", in_h=64, in_w=64)
    ck, per_buf = _synthetic_qat_checkpoint(pkg, sd7b, g)
    weights, quant = statedict.from_qat_checkpoint(ck)
    assert len(quant["input_amax"]) == 67 and len(quant["weight_amax"]) == 67       # 74 convs - 6 plain head outputs - the stem (its own op kind, not listed by the helper)
    b8 = export.export_qat_checkpoint(ck, str(tmp_path / "qat.une"), in_h=64, in_w=64)
    assert b8.precision == export.INT8
    for name, h, w, c, dtype, flags, scale in b8.buffers:
        if dtype == export.BUF_I8 and not name.endswith(".q8"):
            assert scale == pytest.approx(per_buf[name] / 127.0, rel=1e-6), name
    for op, q in zip(b8.ops, b8.op_int8):
        if op.kind == export.OP_CONV:
            carve = any(sg.module.startswith(c) for sg in op.segs for c in ("stem", "stage1_conv", "head_p2"))
            final = any(sg.module.endswith(".2") for sg in op.segs)
            narrow = any(sg.module.startswith(c) for sg in op.segs for c in ("stage1_c3k2.", "fpn_c3k2_3.", "stage2_conv"))   # fp16 by builder choice (export.py _quantize_pass)
            assert q == (not carve and not final and not narrow and op.cin % 64 == 0), op.name
    x = pkg.rng.frame(1234, 64, 64)
    o8, _ = run_op_table(b8, x)
    ref = oracle_mod.forward(osd7b, x, variant="B")
    for n in pkg.graph.OUTPUT_NAMES:
        err = float(np.sqrt(((o8[n] - ref[n]) ** 2).mean()))
        assert err < 0.15 * max(float(ref[n].std()), 0.3), (n, err)
    # a float checkpoint of the QAT topology (no quantizer entries) becomes an fp16 engine
    assert export.export_qat_checkpoint(dict(sd7b), str(tmp_path / "f.une"), in_h=64, in_w=64).precision == export.FP16
