"""GPU parity for graph (B), the reference's QAT topology (unina_yolo_dla/qat.py:350-491): the same engine, kernels
and fusion run a second layer table whose checkpoints come with qat.py's key names. Checked against the CPU oracle
and against fixtures made with the reference qat.py itself (tests/golden/make_golden_qat.py)."""
import numpy as np
import pytest

from conftest import load_golden
from detcmp import compare

pytestmark = pytest.mark.gpu

HEAD_ATOL = 3.0e-2                 # fp16 format noise on logits of std 2.0 behind up to 52 convs (graph (A): 2.5e-2 behind 45)
CLS_RMS, REG_RMS = 7e-3, 2e-3


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


@pytest.fixture(scope="module")
def sd7b(pkg):
    return pkg.synth.make_state_dict(7, pkg.graph.Graph(variant="B"))


@pytest.fixture(scope="module")
def eng640b(pkg, sd7b, torch_cuda):
    from unina_yolo_dla_amd.engine import Engine
    e = Engine.from_state_dict(sd7b, pkg.graph.Graph(variant="B"))
    yield e
    e.close()


def test_graph_b_heads_vs_reference_fixture(pkg, eng640b, torch_cuda):
    gold = load_golden("qat_frame640_seed1234.npz")
    x = torch_cuda.from_numpy(pkg.rng.frame(1234, 640, 640)).cuda()
    heads = eng640b.forward(x)
    assert eng640b.L.unina_fusion_groups(eng640b.h) == 9       # 8 C3k2 blocks + the P2 head run fused
    for name in pkg.graph.OUTPUT_NAMES:
        err = heads[name] - gold[f"head/{name}"]
        assert np.abs(err).max() < HEAD_ATOL, (name, float(np.abs(err).max()))
        assert np.sqrt((err ** 2).mean()) < (CLS_RMS if name.endswith("cls") else REG_RMS), name


def test_graph_b_infer_vs_reference_fixture_detections(pkg, eng640b, torch_cuda):
    gold = load_golden("qat_frame640_seed1234.npz")
    thr = float(gold["conf_thr"])
    x = torch_cuda.from_numpy(pkg.rng.frame(1234, 640, 640)).cuda()
    for q in (0.1, 0.0):
        got = eng640b.infer(x, thr, 0.45, q)
        ref = gold[f"ref_dets_q{q}"]
        want = np.zeros(len(ref), dtype=got.dtype)
        for f in ref.dtype.names:
            want[f] = ref[f]
        stats = compare(got, want, thr, min_iou=0.9983, score_tol=3.5e-3)
        assert stats["matched"] >= 0.97 * len(want), stats
        assert stats["median_dscore"] < 1e-3 and stats["frac_iou_ge_0.999"] >= 0.95, stats


@pytest.mark.parametrize("size", [64, 640])
def test_graph_b_fusion_is_bit_identical(pkg, sd7b, torch_cuda, size):
    from unina_yolo_dla_amd.engine import Engine
    e = Engine.from_state_dict(sd7b, pkg.graph.Graph(variant="B", in_h=size, in_w=size))
    try:
        x = torch_cuda.from_numpy(pkg.rng.frame(1234, size, size)).cuda()
        fused = {k: v.copy() for k, v in e.forward(x).items()}
        bufs = ("cat_fpn3", "cat_fpn2", "cat_fpn1", "cat_pan1", "cat_pan2", "p2_fused", "p3_out", "p4_out")
        fused_bufs = {b: e.read_buffer(b) for b in bufs}
        assert e.set_fusion(False) == 0
        plain = e.forward(x)
        from test_gpu_parity import written     # (down1 / down2 run inside the PAN blocks: their half of the concat is not written)
        for b in bufs:
            assert np.array_equal(written(b, fused_bufs[b]), written(b, e.read_buffer(b))), b
        from test_gpu_parity import same_head
        for k in plain:
            same_head(fused[k], plain[k], k)
    finally:
        e.close()


def test_graph_b_mini64_buffers_vs_oracle(pkg, sd7b, oracle_mod, torch_cuda):
    from unina_yolo_dla_amd.engine import Engine
    g = pkg.graph.Graph(variant="B", in_h=64, in_w=64)
    e = Engine.from_state_dict(sd7b, g)
    osd = oracle_mod.StateDict(sd7b)
    try:
        e.set_fusion(False)
        x = pkg.rng.frame(1234, 64, 64)
        heads = e.forward(torch_cuda.from_numpy(x).cuda())
        ref = oracle_mod.forward(osd, x, keep_all=True, variant="B")
        for name in pkg.graph.OUTPUT_NAMES:
            np.testing.assert_allclose(heads[name], ref[name], atol=HEAD_ATOL, rtol=0, err_msg=name)
        for bname, oname in {"stem": "stem", "stage4_conv": "stage4_conv", "stage4_sppf.cat": "stage4_sppf.cat",
                             "stage4_sppf": "stage4_sppf.cv2", "cat_fpn1": "cat_fpn1", "cat_fpn3": "cat_fpn3",
                             "cat_pan2": "cat_pan2", "p3_out": "pan_c3k2_1.cv3"}.items():
            got, want = e.read_buffer(bname), ref[oname]
            assert got.shape == want.shape, (bname, got.shape, want.shape)
            np.testing.assert_allclose(got, want, atol=1e-2 * max(1.0, float(np.abs(want).max())), rtol=0, err_msg=bname)
    finally:
        e.close()
        osd.close()


def test_graph_b_int8_engine_from_a_qat_checkpoint(pkg, sd7b, oracle_mod, torch_cuda, tmp_path):
    """The QAT-checkpoint path end to end on the GPU: checkpoint (weights + quantizer ranges, qat.py key names) ->
    INT8 engine file of graph (B) -> HIP int8 kernels, against the torch-CPU integer emulation of the same table (codes
    agree up to rare +-1 flips) and, as calibrated drift, against the fp32 oracle."""
    from emulate import run_op_table, engine_buffers, per_op_mismatch
    from test_graph_b_cpu import _synthetic_qat_checkpoint
    from unina_yolo_dla_amd import export
    from unina_yolo_dla_amd.engine import Engine
    g = pkg.graph.Graph(variant="B", in_h=128, in_w=128)
    ck, _ = _synthetic_qat_checkpoint(pkg, sd7b, g)
    path = str(tmp_path / "qat_b.une")
    b8 = export.export_qat_checkpoint(ck, path, in_h=128, in_w=128)
    assert b8.precision == export.INT8 and sum(b8.op_int8) >= 40
    e = Engine(path)
    osd = oracle_mod.StateDict(sd7b)
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
        ref = oracle_mod.forward(osd, x, variant="B")
        for n in pkg.graph.OUTPUT_NAMES:
            # free-running, the engine and the emulation are two equally valid roundings of the same arithmetic: each
            # must sit at the same distance (the quantisation drift) from the fp32 oracle
            e_engine = float(np.sqrt(((heads[n] - ref[n]) ** 2).mean()))
            e_quant = float(np.sqrt(((emu[n] - ref[n]) ** 2).mean()))
            assert e_engine < 1.25 * e_quant + 1e-3, (n, e_engine, e_quant)
            assert e_quant < 0.15 * max(float(ref[n].std()), 0.3), (n, e_quant)
        dets = e.infer(torch_cuda.from_numpy(x).cuda(), 0.75, 0.45, 0.1)
        assert dets.dtype.itemsize == 32 and np.all(np.diff(dets["confidence"]) <= 0)
    finally:
        e.close()
        osd.close()
