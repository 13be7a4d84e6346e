#!/usr/bin/env python3
"""Generates the committed golden fixtures by running the REFERENCE itself in the dev container.

    python tests/golden/make_golden.py            (needs /root/reference; never runs on the GPU box)

What is pinned, and by what:
  * forward tensors   -- /root/reference/unina_yolo_dla/model.py imported as-is (torch CPU fp32),
                         loaded with the build's seeded synthetic state_dict (synth.py);
  * decode/NMS        -- the reference's postprocess.hpp compiled into oracle/_ref (oracle/Makefile);
  * head calibration  -- synth_calib.json (per-head multipliers) measured with the reference model.

Outputs (all small, data only):
  unina-yolo-dla_amd/synth_calib.json
  tests/golden/mini64_seed1234.npz      every module output of a 64x64 forward (fp32 heads, fp16 rest)
  tests/golden/frame640_seed1234.npz    all six heads in full fp32, per-module checksums,
                                        reference detections (postprocess.hpp semantics)
  tests/golden/frame1280_seed1234.npz   per-head checksums + sampled elements at 1280x1280
  tests/golden/lite_p2_64_seed1234.npz  lite_p2=True variant, 64x64 heads
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

import model as ref_model  # noqa: E402  (the reference)
import unina_yolo_dla_amd as u  # noqa: E402
from oracle import oracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 7
FRAME_SEED = 1234
SAMPLES = 16


def ref_net(sd, g):
    m = ref_model.UNINA_YOLO_DLA(g.num_classes, g.base_channels, g.lite_p2).eval()
    res = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and all("num_batches_tracked" in k for k in res.missing_keys), res
    return m


def ref_forward(sd, g, x, hooks=False):
    m = ref_net(sd, g)
    taps = {}
    hs = []
    if hooks:
        for name, mod in m.named_modules():
            if isinstance(mod, (ref_model.ConvBlock, ref_model.Upsample, ref_model.Bottleneck)):
                hs.append(mod.register_forward_hook(
                    lambda _m, _i, o, name=name: taps.__setitem__(name, o[0].numpy().copy())))
        # the three chained pools inside SPPF share one module -> record by call order
        order = []
        m.backbone.sppf.pool.register_forward_hook(lambda _m, _i, o: order.append(o[0].numpy().copy()))
    with torch.no_grad():
        out = m(torch.from_numpy(x))
    for h in hs:
        h.remove()
    heads = {n: t.numpy()[0].copy() for n, t in zip(u.graph.OUTPUT_NAMES, [t for pair in out for t in pair])}
    if hooks:
        for i, t in enumerate(order):
            taps[f"backbone.sppf.pool{i + 1}"] = t
    return heads, taps


def checksum(t):
    flat = t.reshape(-1).astype(np.float64)
    idx = (np.arange(SAMPLES, dtype=np.int64) * 2654435761 + 12345) % flat.size
    return np.array([flat.mean(), np.abs(flat).max()], dtype=np.float64), idx.astype(np.int64), flat[idx].astype(np.float32)


def main():
    torch.set_num_threads(8)
    # ---- 1. head calibration with the reference model as the probe -----------------------------
    g640 = u.graph.Graph()
    scales = u.synth.calibrate_head_scales(
        WEIGHT_SEED, g640, lambda sd, x: ref_forward(sd, g640, x)[0], probe_seed=FRAME_SEED)
    calib = u.synth.load_calib()
    calib[u.synth.calib_key(WEIGHT_SEED, g640)] = scales
    glite = u.graph.Graph(lite_p2=True)
    calib[u.synth.calib_key(WEIGHT_SEED, glite)] = u.synth.calibrate_head_scales(
        WEIGHT_SEED, glite, lambda sd, x: ref_forward(sd, glite, x)[0], probe_seed=FRAME_SEED)
    with open(os.path.join(ROOT, "unina-yolo-dla_amd", "synth_calib.json"), "w") as f:
        json.dump(calib, f, indent=1, sort_keys=True)
    print("head scales:", scales)

    sd = u.synth.make_state_dict(WEIGHT_SEED, g640)

    # ---- 2. 64x64 miniature, every module output ------------------------------------------------
    g64 = u.graph.Graph(in_h=64, in_w=64)
    x64 = u.rng.frame(FRAME_SEED, 64, 64)
    heads, taps = ref_forward(sd, g64, x64, hooks=True)
    blob = {f"head/{k}": v.astype(np.float32) for k, v in heads.items()}
    blob.update({f"tap/{k}": v.astype(np.float16) for k, v in taps.items()})
    np.savez_compressed(os.path.join(GOLD, "mini64_seed1234.npz"), **blob)
    print("mini64:", len(taps), "taps")

    # ---- 3. 640x640: heads + checksums + reference detections ----------------------------------
    x = u.rng.frame(FRAME_SEED, 640, 640)
    heads, taps = ref_forward(sd, g640, x, hooks=True)
    blob = {}
    for k, v in heads.items():
        blob[f"head/{k}"] = v.astype(np.float32)   # full fp32: the decode/NMS fixtures below are exact functions of these
    names = sorted(taps)
    blob["tap_names"] = np.array(names)
    blob["tap_stats"] = np.stack([checksum(taps[n])[0] for n in names])
    blob["tap_idx"] = np.stack([checksum(taps[n])[1] for n in names])
    blob["tap_vals"] = np.stack([checksum(taps[n])[2] for n in names])
    hl = [heads[n] for n in u.graph.OUTPUT_NAMES]
    for q in (0.1, 0.0):
        dets, ncand = oracle.ref_postprocess(hl, 0.5, 0.45, q)
        blob[f"ref_dets_q{q}"] = dets
        blob[f"ref_ncand_q{q}"] = np.array(ncand)
        print(f"640 q={q}: candidates {ncand}, kept {len(dets)}")
    for n in u.graph.OUTPUT_NAMES[::2]:
        c = heads[n]
        conf = 1 / (1 + np.exp(-c.max(axis=0)))
        print(f"  {n}: cells>=0.5: {(conf >= 0.5).sum()}  logits std {c.std():.3f}; reg std {heads[n[:2] + '_reg'].std():.3f}")
    np.savez_compressed(os.path.join(GOLD, "frame640_seed1234.npz"), **blob)

    # ---- 4. 1280x1280: checksums only -----------------------------------------------------------
    g1280 = u.graph.Graph(in_h=1280, in_w=1280)
    x = u.rng.frame(FRAME_SEED, 1280, 1280)
    heads, _ = ref_forward(sd, g1280, x)
    blob = {}
    for k, v in heads.items():
        st, idx, vals = checksum(v)
        blob[f"stats/{k}"], blob[f"idx/{k}"], blob[f"vals/{k}"] = st, idx, vals
    hl = [heads[n] for n in u.graph.OUTPUT_NAMES]
    dets, ncand = oracle.ref_postprocess(hl, 0.6, 0.45, 0.1)
    blob["ref_dets_conf0.6_q0.1"] = dets
    blob["ref_ncand"] = np.array(ncand)
    blob["head/p4_cls"] = heads["p4_cls"].astype(np.float32)
    blob["head/p4_reg"] = heads["p4_reg"].astype(np.float32)
    print(f"1280 conf=0.6: candidates {ncand}, kept {len(dets)}")
    np.savez_compressed(os.path.join(GOLD, "frame1280_seed1234.npz"), **blob)

    # ---- 5. lite_p2 variant ---------------------------------------------------------------------
    gl64 = u.graph.Graph(lite_p2=True, in_h=64, in_w=64)
    sdl = u.synth.make_state_dict(WEIGHT_SEED, gl64)
    heads, _ = ref_forward(sdl, gl64, x64)
    np.savez_compressed(os.path.join(GOLD, "lite_p2_64_seed1234.npz"),
                        **{f"head/{k}": v.astype(np.float32) for k, v in heads.items()})

    # ---- 6. the reference's only known-answer vector (data_loader.py:427-436) -------------------
    # GT [cls,xc,yc,w,h] = [0,.5,.5,.01,.02]; pred [xc,yc,w,h,conf,cls] = [.51,.51,.012,.022,.95,0]; image 640.
    # IoU = 0.0243 < 0.5 and the GT is 6.4x12.8 px (<15 px): expected TP 0 / FP 1 / FN 1.
    with open(os.path.join(GOLD, "small_object_metric_example.json"), "w") as f:
        json.dump({"source": "unina_yolo_dla/data_loader.py:427-436",
                   "image_size": 640, "small_threshold_px": 15, "iou_threshold": 0.5,
                   "preds": [[0.51, 0.51, 0.012, 0.022, 0.95, 0]],
                   "targets": [[0, 0.5, 0.5, 0.01, 0.02], [1, 0.2, 0.3, 0.05, 0.08]],
                   "expected": {"tp": 0, "fp": 1, "fn": 1, "iou": 0.0243}}, f, indent=1)
    print("done")


if __name__ == "__main__":
    main()
