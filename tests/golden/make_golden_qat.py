#!/usr/bin/env python3
"""Golden fixtures for graph (B): the reference's QAT model ``UNINA_YOLO_DLA_QAT`` (unina_yolo_dla/qat.py:350-491),
run by importing the reference's qat.py as-is in the dev container.

    python tests/golden/make_golden_qat.py        (needs /root/reference; never runs on the GPU box)

pytorch-quantization is not installed here, so the reference's own fallback is what runs: QuantConvBlock wraps a plain
nn.Conv2d (qat.py:249-254) and QuantBottleneck adds the shortcut unquantised (qat.py:290-292) -- i.e. the float forward
of the QAT TOPOLOGY. That pins the layer table / key names / concat orders of graph (B); the fake-quant arithmetic
itself stays "parity unpinned" (DESIGN.md section 2).

Outputs (data only):
  unina-yolo-dla_amd/synth_calib.json      + the graph-(B) head multipliers (key ..._B)
  tests/golden/qat_mini64_seed1234.npz     every module output of a 64x64 forward (fp32 heads, fp16 rest)
  tests/golden/qat_frame640_seed1234.npz   six heads in fp32, reference detections (postprocess.hpp semantics)
"""
import json
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/unina_yolo_dla")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import qat as ref_qat  # noqa: E402  (the reference)
import unina_yolo_dla_amd as u  # noqa: E402
from oracle import oracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 7
FRAME_SEED = 1234
CONF_THR = 0.75   # the seeded graph-(B) logits are not zero-mean before the bias: 0.5 would pass 5 615 cells (> MAX_DETECTIONS)


def ref_forward(sd, g, x, hooks=False):
    m = ref_qat.UNINA_YOLO_DLA_QAT(g.num_classes, g.base_channels).eval()
    res = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and all("num_batches_tracked" in k for k in res.missing_keys), res
    taps, hs, order = {}, [], []
    if hooks:
        for name, mod in m.named_modules():
            if isinstance(mod, (ref_qat.QuantConvBlock, ref_qat.QuantBottleneck)):
                hs.append(mod.register_forward_hook(
                    lambda _m, _i, o, name=name: taps.__setitem__(name, o[0].numpy().copy())))
        m.stage4_sppf.pool.register_forward_hook(lambda _m, _i, o: order.append(o[0].numpy().copy()))
    with torch.no_grad():
        out = m(torch.from_numpy(x))
    for h in hs:
        h.remove()
    heads = {n: t.numpy()[0].copy() for n, t in zip(u.graph.OUTPUT_NAMES, [t for pair in out for t in pair])}
    for i, t in enumerate(order):
        taps[f"stage4_sppf.pool{i + 1}"] = t
    return heads, taps


def main():
    assert not ref_qat.QUANT_AVAILABLE, "fixtures are defined for the reference's float fallback"
    torch.set_num_threads(8)
    g640 = u.graph.Graph(variant="B")
    scales = u.synth.calibrate_head_scales(WEIGHT_SEED, g640, lambda sd, x: ref_forward(sd, g640, x)[0], probe_seed=FRAME_SEED)
    calib = u.synth.load_calib()
    calib[u.synth.calib_key(WEIGHT_SEED, g640)] = scales
    with open(os.path.join(ROOT, "unina-yolo-dla_amd", "synth_calib.json"), "w") as f:
        json.dump(calib, f, indent=1, sort_keys=True)
    print("head scales (B):", scales)
    sd = u.synth.make_state_dict(WEIGHT_SEED, g640)
    n_params = sum(p.numel() for p in ref_qat.UNINA_YOLO_DLA_QAT().parameters())
    print("reference parameter count:", n_params)

    g64 = u.graph.Graph(variant="B", in_h=64, in_w=64)
    x64 = u.rng.frame(FRAME_SEED, 64, 64)
    heads, taps = ref_forward(sd, g64, x64, hooks=True)
    blob = {f"head/{k}": v.astype(np.float32) for k, v in heads.items()}
    blob.update({f"tap/{k}": v.astype(np.float16) for k, v in taps.items()})
    blob["n_params"] = np.array(n_params)
    np.savez_compressed(os.path.join(GOLD, "qat_mini64_seed1234.npz"), **blob)
    print("qat mini64:", len(taps), "taps")

    x = u.rng.frame(FRAME_SEED, 640, 640)
    heads, _ = ref_forward(sd, g640, x)
    blob = {f"head/{k}": v.astype(np.float32) for k, v in heads.items()}
    hl = [heads[n] for n in u.graph.OUTPUT_NAMES]
    blob["conf_thr"] = np.array(CONF_THR)
    for q in (0.1, 0.0):
        dets, ncand = oracle.ref_postprocess(hl, CONF_THR, 0.45, q)
        blob[f"ref_dets_q{q}"] = dets
        blob[f"ref_ncand_q{q}"] = np.array(ncand)
        print(f"640 q={q}: candidates {ncand}, kept {len(dets)}")
    np.savez_compressed(os.path.join(GOLD, "qat_frame640_seed1234.npz"), **blob)
    print("done")


if __name__ == "__main__":
    main()
