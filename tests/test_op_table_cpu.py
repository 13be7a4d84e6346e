"""The exporter's fused op table, executed by the test-only torch emulator (tests/emulate.py) in fp32, must
reproduce the fp32 oracle: proves BN folding, sibling-conv merging, concat slicing, residual and upsample folding,
SPPF fusion -- with no GPU involved."""
import numpy as np
import torch

from emulate import run_op_table


def test_op_table_fp32_equals_oracle_mini64(pkg, sd7, oracle_mod, oracle_sd7):
    from unina_yolo_dla_amd import export
    torch.set_num_threads(4)
    g = pkg.graph.Graph(in_h=64, in_w=64)
    b = export.EngineBuilder(sd7, g)
    x = pkg.rng.frame(1234, 64, 64)
    # the emulator reads fp16-rounded folded weights from the blob: compare against the oracle with a tolerance that
    # covers weight rounding only (activations stay fp32 here)
    outs, named = run_op_table(b, x, fp16=False)
    ref = oracle_mod.forward(oracle_sd7, x, keep_all=True)
    for n in pkg.graph.OUTPUT_NAMES:
        np.testing.assert_allclose(outs[n], ref[n], atol=1.5e-2, rtol=0, err_msg=n)
        assert np.sqrt(((outs[n] - ref[n]) ** 2).mean()) < 4e-3
    for bname, oname in {"backbone.stem": "backbone.stem", "neck.cat_fpn1": "neck.cat_fpn1",
                         "neck.cat_pan2": "neck.cat_pan2", "backbone.sppf.cat": "backbone.sppf.cat",
                         "p2_fused": "neck.fpn_c3k2_2.cv3"}.items():
        np.testing.assert_allclose(named[bname], ref[oname], atol=1e-2 * max(1.0, np.abs(ref[oname]).max()), rtol=0, err_msg=bname)


def test_lite_p2_op_table(pkg, oracle_mod):
    from unina_yolo_dla_amd import export
    g = pkg.graph.Graph(lite_p2=True, in_h=64, in_w=64)
    sd = pkg.synth.make_state_dict(7, g)
    b = export.EngineBuilder(sd, g)
    assert len(b.ops) == 52 - 3                      # stage1 C3k2 (4 launches) -> one 3x3 conv
    x = pkg.rng.frame(1234, 64, 64)
    outs, _ = run_op_table(b, x, fp16=False)
    osd = oracle_mod.StateDict(sd)
    ref = oracle_mod.forward(osd, x, lite_p2=True)
    osd.close()
    for n in pkg.graph.OUTPUT_NAMES:
        np.testing.assert_allclose(outs[n], ref[n], atol=1.5e-2, rtol=0, err_msg=n)
