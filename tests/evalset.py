"""Synthetic YOLO-layout dataset for the evaluation rows: frames are the seeded network
tensors (.npy), labels are derived from the ORACLE's detections of each frame (every second detection, jittered by a
fixed pattern, written as 'cls xc yc w h' normalised rows) so that the small-object metric and the conformal
calibration see true positives, false positives and false negatives. Test infrastructure only."""
import os

import numpy as np


class OracleDetector:
    """frame -> detections through the CPU oracle (the checker), same signature as evaluate.EngineDetector."""

    def __init__(self, oracle_mod, osd, names, **fwd_kw):
        self.o, self.osd, self.names, self.kw = oracle_mod, osd, names, fwd_kw

    def __call__(self, frame, conf, iou, q):
        x = np.ascontiguousarray(frame, dtype=np.float32)[None]
        heads = self.o.forward(self.osd, x, **self.kw)
        dets, _ = self.o.postprocess([heads[n] for n in self.names], conf, iou, q)
        return dets


def make_dataset(root, pkg, detect, size, seeds, conf=0.5):
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    os.makedirs(os.path.join(root, "labels"), exist_ok=True)
    n_labels = 0
    for seed in seeds:
        x = pkg.rng.frame(seed, size, size)[0]
        np.save(os.path.join(root, "images", f"frame{seed}.npy"), x)
        dets = detect(x, conf, 0.45, 0.0)
        rows = []
        for i, d in enumerate(dets[::2]):
            # fixed jitter pattern: shifts of 0 .. 0.75 px and a 0 .. 6 % size change -> IoU with the detection 0.8 .. 1.0
            dx, dy = 0.25 * (i % 4), 0.25 * ((i // 4) % 4)
            s = 0.85 if i % 3 == 0 else 1.0 + 0.02 * (i % 4) - 0.03     # every third label shrunk: under the 15 px 'small' limit, IoU ~0.7
            w, h = (d["x2"] - d["x1"]) * s, (d["y2"] - d["y1"]) * s
            xc, yc = (d["x1"] + d["x2"]) / 2 + dx, (d["y1"] + d["y2"]) / 2 + dy
            rows.append(f"{int(d['class_id'])} {xc / size:.6f} {yc / size:.6f} {w / size:.6f} {h / size:.6f}")
        # two ground truths nothing detects (false negatives), one of them small
        rows.append(f"0 0.031250 0.031250 {10 / size:.6f} {12 / size:.6f}")
        rows.append(f"1 0.500000 0.970000 {40 / size:.6f} {40 / size:.6f}")
        with open(os.path.join(root, "labels", f"frame{seed}.txt"), "w") as f:
            f.write("\n".join(rows) + "\n")
        n_labels += len(rows)
    return n_labels
